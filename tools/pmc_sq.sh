# SQ counters of the in-loop NN kernel (own pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_sq
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES -d $GRAFT_REPO_ROOT/gpurun_out/prof_sq -o q -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --iters ${ITERS:-60} --frames ${FRAMES:-1024} --no-cpu-baseline --no-logging-run > $GRAFT_REPO_ROOT/gpurun_out/prof_sq.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof_sq/q_results.db | grep -E "nn_stream|nn_mfma" | head -3; python3 $GRAFT_REPO_ROOT/tools/pmc_kernel.py $GRAFT_REPO_ROOT/gpurun_out/prof_sq/q_results.db nn_stream4 ${SKIP:-60}
