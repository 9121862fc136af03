set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_issue_probe tools/valu_issue_probe.hip 2>/dev/null && /tmp/valu_issue_probe > $O/valu_issue_probe.txt 2>&1
hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_repro tools/pk_f32_mfma_repro.hip 2>/dev/null && timeout 600 /tmp/pk_repro 40 > $O/pk_repro.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_general_k.py tests/test_gpu_sharded.py -x -q > $O/t_second.log 2>&1; echo second rc=$?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/avail_sq.txt
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --value-only"
rm -rf /tmp/prof_r4_sq2
timeout 900 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES -d /tmp/prof_r4_sq2 -o m -- $B > $O/prof_r4_sq2.log 2>&1
cd $GRAFT_REPO_ROOT
for k in nn_stream4 panel_gemm3_rb2_kernel panel_gemm3_rb2k vposer_fwd vposer_bwd pose_fwd pose_bwd skin_fwd skin_bwd adam_step; do python tools/pmc_kernel.py /tmp/prof_r4_sq2/m_results.db $k 0; done > $O/r4_pmc_SQ2_per_kernel.txt
python tools/pmc_kernel.py /tmp/prof_r4_sq2/m_results.db nn_stream4 300 > $O/r4_pmc_SQ2_nn_in_loop_steady.txt
cat $O/valu_issue_probe.txt $O/pk_repro.txt $O/r4_pmc_SQ2_nn_in_loop_steady.txt; tail -5 $O/t_second.log; tail -3 $O/prof_r4_sq2.log
