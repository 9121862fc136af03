set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_repro tools/pk_f32_mfma_repro.hip 2>/dev/null && timeout 900 /tmp/pk_repro 40 > $O/pk_repro.txt 2>&1
cat $O/pk_repro.txt
hipcc --offload-arch=gfx950 -O3 -DPK_TREE_NOPK -o /tmp/pk_repro_nopk tools/pk_f32_mfma_repro.hip 2>$O/pk_repro_nopk_build.log && timeout 900 /tmp/pk_repro_nopk 40 > $O/pk_repro_nopk.txt 2>&1
cat $O/pk_repro_nopk.txt | tail -5
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
FDC_PK=+ bash tools/build_variant.sh pk > $O/build_pk.log 2>&1
timeout 900 python tools/pk_bisect.py 300 pk > $O/pk_bisect.txt 2>&1
cat $O/pk_bisect.txt | grep -v Warning
