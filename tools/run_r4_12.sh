set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
FDC_PK=+ bash tools/build_variant.sh pkdbg -DFDC_DEBUG_BUFFERS > $O/build_pkdbg.log 2>&1 || tail -5 $O/build_pkdbg.log
timeout 600 python tools/pk_where.py 300 pkdbg 2>&1 | grep -v Warning | cut -c1-700
hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_repro tools/pk_f32_mfma_repro.hip 2>/dev/null && timeout 900 /tmp/pk_repro 40 2>&1 | tail -4
