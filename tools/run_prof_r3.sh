# Round-3 measurement set on the GPU box: bench line + rocprofv3 kernel trace + PMC passes (each its own pass, kernel-trace only).
# Summaries for profiles/ are written to gpurun_out/profiles_r3/ (copy them into profiles/ and commit).
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
P=$OUT/profiles_r3
mkdir -p $P
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-logging-run --no-exact-fp32"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r3*
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r3 -o r3 -- $B > $OUT/prof_r3_bench.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_r3_fetch -o f -- $B > $OUT/prof_r3_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_r3_write -o w -- $B > $OUT/prof_r3_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_r3_sq -o m -- $B > $OUT/prof_r3_sq.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py /tmp/prof_r3/r3_results.db $P/r3_kernel_trace_stats_bench_500it.txt > /dev/null
python tools/rocpd_series.py /tmp/prof_r3/r3_results.db nn_stream4 50 > $P/r3_nn_in_loop_duration_series.txt
python tools/rocpd_summary.py /tmp/prof_r3_fetch/f_results.db $P/r3_pmc_FETCH_SIZE_bench_500it.txt > /dev/null
python tools/rocpd_summary.py /tmp/prof_r3_write/w_results.db $P/r3_pmc_WRITE_SIZE_bench_500it.txt > /dev/null
python tools/rocpd_summary.py /tmp/prof_r3_sq/m_results.db $P/r3_pmc_SQ_bench_500it.txt > /dev/null
python tools/make_pmc_json.py /tmp/prof_r3/r3_results.db /tmp/prof_r3_fetch/f_results.db /tmp/prof_r3_write/w_results.db /tmp/prof_r3_sq/m_results.db $P/r3_pmc_summary.json 300 > /dev/null
for k in nn_stream4 panel_gemm3_rb2_kernel panel_gemm3_rb2k panel_gemm3_wide vposer_fwd vposer_bwd pose_fwd pose_bwd skin_fwd skin_bwd adam_step; do python tools/pmc_kernel.py /tmp/prof_r3_sq/m_results.db $k 0; done > $P/r3_pmc_SQ_per_kernel.txt
python tools/pmc_kernel.py /tmp/prof_r3_sq/m_results.db nn_stream4 300 > $P/r3_pmc_SQ_nn_in_loop_steady.txt
python tools/pmc_kernel.py /tmp/prof_r3_sq/m_results.db nn_mfma_kernel 0 > $P/r3_pmc_SQ_nn_bruteforce.txt
head -16 $P/r3_kernel_trace_stats_bench_500it.txt; cat $P/r3_pmc_SQ_nn_in_loop_steady.txt; ls $P
