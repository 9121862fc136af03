# Round-1 measurement set on the GPU box: tests, smoke, bench line, rocprofv3 kernel trace + PMC passes.
# Summaries for profiles/ are written to gpurun_out/profiles_r1/ (copy them into profiles/ and commit).
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT/profiles_r1
python -m pytest tests -m gpu -q 2>&1 | tail -2
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py > $OUT/bench_r1_full.log 2>&1
tail -1 $OUT/bench_r1_full.log > $OUT/profiles_r1/r1_bench_1gpu.json
tail -1 $OUT/bench_r1_full.log | python tools/bench_line.py full
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_r1*
rocprofv3 --kernel-trace --stats -d $OUT/prof_r1 -o r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/prof_r1_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_r1_fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --iters 60 --no-cpu-baseline > $OUT/prof_r1_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_r1_write -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --iters 60 --no-cpu-baseline > $OUT/prof_r1_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY -d $OUT/prof_r1_mfma -o m -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --iters 60 --no-cpu-baseline > $OUT/prof_r1_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $OUT/prof_r1/r1_results.db $OUT/profiles_r1/r1_kernel_trace_stats_bench_500it.txt > /dev/null
python tools/rocpd_summary.py $OUT/prof_r1_fetch/f_results.db $OUT/profiles_r1/r1_pmc_FETCH_SIZE_bench_60it.txt > /dev/null
python tools/rocpd_summary.py $OUT/prof_r1_write/w_results.db $OUT/profiles_r1/r1_pmc_WRITE_SIZE_bench_60it.txt > /dev/null
python tools/rocpd_summary.py $OUT/prof_r1_mfma/m_results.db $OUT/profiles_r1/r1_pmc_SQ_bench_60it.txt > /dev/null
python tools/make_traffic_json.py $OUT/prof_r1_fetch/f_results.db $OUT/prof_r1_write/w_results.db $OUT/profiles_r1/r1_pmc_traffic.json 30
python tools/pmc_kernel.py $OUT/prof_r1_mfma/m_results.db nn_stream4 30 | tee $OUT/profiles_r1/r1_pmc_SQ_nn_in_loop_steady.txt
python tools/pmc_kernel.py $OUT/prof_r1_mfma/m_results.db nn_mfma_kernel 0 | tee $OUT/profiles_r1/r1_pmc_SQ_nn_bruteforce.txt
ls $OUT/profiles_r1
