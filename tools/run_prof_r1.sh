set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -2
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py > gpurun_out/bench_r1_full.log 2>&1
tail -1 gpurun_out/bench_r1_full.log | python tools/bench_line.py full
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r1*
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1_fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --iters 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1_write -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --iters 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1_mfma -o m -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --iters 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1_mfma.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/
