set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_issue_probe tools/valu_issue_probe.hip 2>/dev/null && /tmp/valu_issue_probe > $O/valu_issue_probe.txt 2>&1
cat $O/valu_issue_probe.txt
FDC_PK=+ bash tools/build_variant.sh pk > $O/build_pk.log 2>&1
timeout 900 python tools/pk_bisect.py 300 pk > $O/pk_bisect.txt 2>&1
cat $O/pk_bisect.txt | grep -v Warning
timeout 1200 python -m pytest tests/test_gpu_parity500.py -x -q -s > $O/t_parity500.log 2>&1; echo parity500 rc=$?
grep -v "^$" $O/t_parity500.log | tail -25
timeout 2400 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; echo gpu rc=$?
tail -5 $O/t_gpu.log
