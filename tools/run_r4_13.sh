set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_repro tools/pk_f32_mfma_repro.hip 2>/dev/null && timeout 900 /tmp/pk_repro 40 2>&1 | tail -8
FDC_PK=+ bash tools/build_variant.sh pk > $O/build_pk.log 2>&1
timeout 600 python tools/pk_bisect.py 300 pk 2>&1 | grep "^A\.\|^B\.\|library\|Error" | cut -c1-300
