# Round-2 measurement set on the GPU box: tests, smoke, bench line, rocprofv3 kernel trace + PMC passes (each its own pass,
# kernel-trace only).  Summaries for profiles/ are written to gpurun_out/profiles_r2/ (copy them into profiles/ and commit).
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
P=$OUT/profiles_r2
mkdir -p $P
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py > $OUT/bench_r2_full.log 2>&1
tail -1 $OUT/bench_r2_full.log > $P/r2_bench_1gpu.json
tail -1 $OUT/bench_r2_full.log | python tools/bench_line.py full
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r2*
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r2 -o r2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-logging-run > $OUT/prof_r2_bench.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_r2_fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-logging-run > $OUT/prof_r2_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_r2_write -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-logging-run > $OUT/prof_r2_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_r2_mfma -o m -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-logging-run > $OUT/prof_r2_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py /tmp/prof_r2/r2_results.db $P/r2_kernel_trace_stats_bench_500it.txt > /dev/null
python tools/rocpd_series.py /tmp/prof_r2/r2_results.db nn_stream4 50 > $P/r2_nn_in_loop_duration_series.txt
python tools/rocpd_summary.py /tmp/prof_r2_fetch/f_results.db $P/r2_pmc_FETCH_SIZE_bench_500it.txt > /dev/null
python tools/rocpd_summary.py /tmp/prof_r2_write/w_results.db $P/r2_pmc_WRITE_SIZE_bench_500it.txt > /dev/null
python tools/rocpd_summary.py /tmp/prof_r2_mfma/m_results.db $P/r2_pmc_SQ_bench_500it.txt > /dev/null
python tools/make_traffic_json.py /tmp/prof_r2_fetch/f_results.db /tmp/prof_r2_write/w_results.db $P/r2_pmc_traffic.json 300 > /dev/null
for k in nn_stream4 panel_gemm3_rb2_kernel panel_gemm3_rb2k panel_gemm3_wide vposer_fwd vposer_bwd pose_fwd pose_bwd skin_fwd skin_bwd; do python tools/pmc_kernel.py /tmp/prof_r2_mfma/m_results.db $k 0; done > $P/r2_pmc_SQ_per_kernel.txt
python tools/pmc_kernel.py /tmp/prof_r2_mfma/m_results.db nn_stream4 300 > $P/r2_pmc_SQ_nn_in_loop_steady.txt
python tools/pmc_kernel.py /tmp/prof_r2_mfma/m_results.db nn_mfma_kernel 0 > $P/r2_pmc_SQ_nn_bruteforce.txt
head -14 $P/r2_kernel_trace_stats_bench_500it.txt; cat $P/r2_pmc_SQ_per_kernel.txt; ls $P
