# kernel trace of one bench step at a given shard size
cd /tmp && export TMPDIR=/tmp
F=${FRAMES:-1024}
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_t$F
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_t$F -o t -- python3 $GRAFT_REPO_ROOT/bench.py --frames $F --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_t$F.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof_t$F/t_results.db | head -32
