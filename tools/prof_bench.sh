# rocprofv3 kernel trace of one bench step -> gpurun_out/<name>_kernels.txt   usage: tools/prof_bench.sh <name> [bench args]
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_$name
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pb_$name -o pb -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-logging-run "$@" > /dev/null 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/pb_$name/pb_results.db $GRAFT_REPO_ROOT/gpurun_out/${name}_kernels.txt | head -24
