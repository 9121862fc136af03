# Round-4 measurement set on the GPU box: bench line + rocprofv3 kernel trace + PMC passes (each its own pass, kernel-trace only).
# Summaries for profiles/ are written to gpurun_out/profiles_r4/ (copy them into profiles/ and commit).
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
P=$OUT/profiles_r4
mkdir -p $P
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-logging-run --no-exact-fp32"
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_issue_probe tools/valu_issue_probe.hip 2>/dev/null && /tmp/valu_issue_probe > $P/r4_valu_issue_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r4*
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r4 -o r4 -- $B > $OUT/prof_r4_bench.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_r4_fetch -o f -- $B > $OUT/prof_r4_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_r4_write -o w -- $B > $OUT/prof_r4_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/prof_r4_sq -o m -- $B > $OUT/prof_r4_sq.log 2>&1
timeout 900 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d /tmp/prof_r4_sq2 -o m -- $B > $OUT/prof_r4_sq2.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d /tmp/prof_r4_mix -o m -- $B > $OUT/prof_r4_mix.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py /tmp/prof_r4/r4_results.db $P/r4_kernel_trace_stats_bench_500it.txt > /dev/null
python tools/rocpd_series.py /tmp/prof_r4/r4_results.db nn_stream4 50 > $P/r4_nn_in_loop_duration_series.txt
python tools/rocpd_summary.py /tmp/prof_r4_fetch/f_results.db $P/r4_pmc_FETCH_SIZE_bench_500it.txt > /dev/null
python tools/rocpd_summary.py /tmp/prof_r4_write/w_results.db $P/r4_pmc_WRITE_SIZE_bench_500it.txt > /dev/null
python tools/rocpd_summary.py /tmp/prof_r4_sq/m_results.db $P/r4_pmc_SQ_bench_500it.txt > /dev/null
python tools/make_pmc_json.py /tmp/prof_r4/r4_results.db /tmp/prof_r4_fetch/f_results.db /tmp/prof_r4_write/w_results.db /tmp/prof_r4_sq/m_results.db $P/r4_pmc_summary.json 300 /tmp/prof_r4_sq2/m_results.db /tmp/prof_r4_mix/m_results.db > /dev/null
for db in sq sq2 mix; do
  for k in nn_stream4 panel_gemm3_rb2_kernel panel_gemm3_rb2k panel_gemm3_wide vposer_fwd vposer_bwd pose_fwd pose_bwd skin_fwd skin_bwd adam_step; do python tools/pmc_kernel.py /tmp/prof_r4_$db/m_results.db $k 0; done > $P/r4_pmc_${db}_per_kernel.txt
  python tools/pmc_kernel.py /tmp/prof_r4_$db/m_results.db nn_stream4 300 > $P/r4_pmc_${db}_nn_in_loop_steady.txt
done
python tools/pmc_kernel.py /tmp/prof_r4_sq/m_results.db nn_mfma_kernel 0 > $P/r4_pmc_sq_nn_bruteforce.txt
head -16 $P/r4_kernel_trace_stats_bench_500it.txt; cat $P/r4_pmc_sq2_nn_in_loop_steady.txt; ls $P
hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_repro tools/pk_f32_mfma_repro.hip 2>/dev/null && timeout 600 /tmp/pk_repro 20 > $P/r4_pk_f32_repro.txt 2>&1
cat $P/r4_pk_f32_repro.txt
