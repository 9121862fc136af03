# kernel times of the panel kernels for the product build and the FDC_PN_ABL ablation builds
cd /tmp && export TMPDIR=/tmp
for v in "" _plain _abl1 _abl2 _abl3; do
  export FDCAP_LIB=$GRAFT_REPO_ROOT/4dcapture-fpv_amd/libfdcap_hip$v.so
  rm -rf /tmp/pp$v
  timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pp$v -o pp -- python3 $GRAFT_REPO_ROOT/tools/panel_probe.py > /dev/null 2>&1
  echo "== variant '$v'"
  python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/pp$v/pp_results.db | grep "panel_gemm\|vposer_fwd\|vposer_bwd"
done
