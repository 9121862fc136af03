set -x
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r1_full.log 2>&1
tail -1 gpurun_out/bench_r1_full.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --iters 50 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1_bench.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_r1 | head -20
