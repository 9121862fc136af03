# first GPU call of round 5: baseline bench line of the round-4 tree on this round's boxes + the three per-config profile sets
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --no-cpu-baseline > gpurun_out/r5_bench_base.json 2> gpurun_out/r5_bench_base.err
tail -c 1500 gpurun_out/r5_bench_base.json
bash tools/run_prof_r5.sh c3 300
bash tools/run_prof_r5.sh c5 40 --frames 512 --scene 2000000 --all-contacts --iters 100
bash tools/run_prof_r5.sh c2 300 --frames 256 --scene 100000
