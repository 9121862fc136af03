set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_issue_probe tools/valu_issue_probe.hip 2>/dev/null && /tmp/valu_issue_probe > $O/valu_issue_probe.txt 2>&1
cat $O/valu_issue_probe.txt
hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_repro tools/pk_f32_mfma_repro.hip 2>/dev/null && timeout 600 /tmp/pk_repro 40 > $O/pk_repro.txt 2>&1
cat $O/pk_repro.txt
FDC_PK=+ bash tools/build_variant.sh pk > $O/build_pk.log 2>&1
timeout 900 python tools/pk_bisect.py 300 pk > $O/pk_bisect.txt 2>&1
cat $O/pk_bisect.txt | grep -v Warning
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --value-only"
rm -rf /tmp/prof_r4_mix
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA -d /tmp/prof_r4_mix -o m -- $B > $O/prof_r4_mix.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_kernel.py /tmp/prof_r4_mix/m_results.db nn_stream4 300 > $O/r4_pmc_mix_nn_in_loop_steady.txt
cat $O/r4_pmc_mix_nn_in_loop_steady.txt
