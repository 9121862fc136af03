cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_sq2
rocprofv3 --pmc ${CTRS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES} -d $GRAFT_REPO_ROOT/gpurun_out/prof_sq2 -o q -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --iters ${ITERS:-100} --frames ${FRAMES:-1024} --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_sq2.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_sq2.log | cut -c1-300
python3 $GRAFT_REPO_ROOT/tools/pmc_kernel.py $GRAFT_REPO_ROOT/gpurun_out/prof_sq2/q_results.db ${KERNEL:-nn_stream4} ${SKIP:-60}
