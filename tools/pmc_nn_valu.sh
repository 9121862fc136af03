# SQ_INSTS_VALU of the in-loop NN kernel over a whole fit, product library or FDCAP_LIB (own pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_nnv
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES -d /tmp/pmc_nnv -o q -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-logging-run --no-exact-fp32 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_kernel.py /tmp/pmc_nnv/q_results.db nn_stream4 ${SKIP:-300}
