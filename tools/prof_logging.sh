# rocprofv3 kernel trace of the every-iteration-logging fit (bench's secondary run): steps 1, warmup 0 + 1 logging step
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_log
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pb_log -o pb -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/pb_log/pb_results.db $GRAFT_REPO_ROOT/gpurun_out/logging_kernels.txt | head -30
