# Round-4 measurement call on the GPU box: build, profile set (tools/run_prof_r4.sh -> gpurun_out/profiles_r4/), the full bench line,
# the whole GPU test suite.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
bash tools/run_prof_r4.sh > $O/run_prof_r4.log 2>&1
tail -40 $O/run_prof_r4.log
python bench.py --steps 20 --warmup 5 > $O/bench_r4_full.json 2> $O/bench_r4_full.err
tail -c 3000 $O/bench_r4_full.json
timeout 2400 python -m pytest tests -m gpu -x -q > $O/t_gpu.log 2>&1; echo gpu rc=$?
tail -4 $O/t_gpu.log
