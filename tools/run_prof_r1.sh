set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -2
python bench.py > gpurun_out/bench_r1_full.log 2>&1
tail -1 gpurun_out/bench_r1_full.log | python tools/bench_line.py full
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r1*
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --iters 50 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1_bench.log 2>&1
