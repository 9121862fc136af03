# round 5, final measurement set on ONE box: the three per-configuration profile sets (kernel trace + PMC passes, 500 iterations each),
# the every-iteration-logging fit's launch series, the brute-force / wide-GEMM counters (r4's command), the bench line itself.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles_r5
bash tools/run_prof_r5.sh c3 300 > gpurun_out/prof_r5_c3.log 2>&1
bash tools/run_prof_r5.sh c5 300 --config c5 > gpurun_out/prof_r5_c5.log 2>&1
bash tools/run_prof_r5.sh c2 300 --config c2 > gpurun_out/prof_r5_c2.log 2>&1
# the logging fit launch by launch (the phase switch)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_r5_log
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r5_log -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --value-only --profile-logging > $GRAFT_REPO_ROOT/gpurun_out/prof_r5_log.log 2>&1
cd $GRAFT_REPO_ROOT
P=gpurun_out/profiles_r5
python - > $P/r5_nn_logging_series.txt <<'PY'
import sqlite3
c = sqlite3.connect("/tmp/prof_r5_log/t_results.db")
d = [(e - s) / 1e3 for s, e in c.execute("select start, end from kernels where name like '%nn_stream4%' order by start")]
print("# tools/run_round5_final.sh: rocprofv3 --kernel-trace of ONE every-iteration-logging fit (bench.py --steps 1 --warmup 0 --value-only --profile-logging):")
print("# the Chamfer search launch by launch -- 400 in phase 1 (used by the loss), 100 in phase 2 (only the printed contact term needs them)")
print(f"nn_stream4 in one logging fit: {len(d)} launches, mean {sum(d)/len(d):.1f} us")
print("per block of 25:", " ".join(f"{sum(d[i:i+25])/len(d[i:i+25]):.1f}" for i in range(0, len(d), 25)))
print("launches 400..:", " ".join(f"{v:.0f}" for v in d[400:]))
PY
cat $P/r5_nn_logging_series.txt | cut -c1-400
# r4's command (brute-force launch + wide GEMM in the same process) for the two kernels the r5 passes do not run: SQ pass only
cd /tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-logging-run --no-exact-fp32 --no-other-configs"
rm -rf /tmp/prof_r5_bf*
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_r5_bf -o t -- $B > /dev/null 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_r5_bf_sq -o m -- $B > /dev/null 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_r5_bf_f -o f -- $B > /dev/null 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_r5_bf_w -o w -- $B > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/make_pmc_json.py /tmp/prof_r5_bf/t_results.db /tmp/prof_r5_bf_f/f_results.db /tmp/prof_r5_bf_w/w_results.db /tmp/prof_r5_bf_sq/m_results.db $P/r5_c3_ops_pmc_summary.json 300 > /dev/null
python bench.py > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err
tail -c 400 gpurun_out/r5_bench_final.json
