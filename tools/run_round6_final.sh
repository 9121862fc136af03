# round 6, measurement set on ONE box: the three per-configuration profile sets (kernel trace + PMC passes, 500 iterations each), the
# operator kernels' counters (brute-force search, wide product: r4's command), set-up timing, per-launch times at the shard sizes,
# the bench line itself.  Summaries land in gpurun_out/profiles_r6/ (copy into profiles/ and commit).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles_r6
P=gpurun_out/profiles_r6
bash tools/run_prof_r6.sh c3 300 > gpurun_out/prof_r6_c3.log 2>&1
bash tools/run_prof_r6.sh c5 300 --config c5 > gpurun_out/prof_r6_c5.log 2>&1
bash tools/run_prof_r6.sh c2 300 --config c2 > gpurun_out/prof_r6_c2.log 2>&1
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-logging-run --no-exact-fp32 --no-other-configs"
rm -rf /tmp/prof_r6_bf*
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_r6_bf -o t -- $B > /dev/null 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_r6_bf_sq -o m -- $B > /dev/null 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_r6_bf_f -o f -- $B > /dev/null 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_r6_bf_w -o w -- $B > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/make_pmc_json.py /tmp/prof_r6_bf/t_results.db /tmp/prof_r6_bf_f/f_results.db /tmp/prof_r6_bf_w/w_results.db /tmp/prof_r6_bf_sq/m_results.db $P/r6_c3_ops_pmc_summary.json 300 > /dev/null
python tools/setup_timing.py c3 c2 c5 cli300 > $P/r6_setup_timing.txt 2>&1
python tools/launch_times.py 1024 512 256 128 2>&1 | grep frames > $P/r6_launch_times.txt
python tools/launch_times.py --config c5 512 256 128 64 2>&1 | grep frames >> $P/r6_launch_times.txt
python bench.py > gpurun_out/r6_bench_final.json 2> gpurun_out/r6_bench_final.err
tail -c 400 gpurun_out/r6_bench_final.json
