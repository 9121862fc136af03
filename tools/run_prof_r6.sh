# Round-6 measurement set on the GPU box for ONE configuration: rocprofv3 kernel trace + PMC passes (each its own pass,
# kernel-trace only).  usage: bash tools/run_prof_r6.sh <tag> <steady-skip> <bench.py size flags ...>
#   tag c3: (no flags)                                  BASELINE config 3 (the quoted one)
#   tag c5: --frames 512 --scene 2000000 --all-contacts  BASELINE config 5 (the named HBM stress)
#   tag c2: --frames 256 --scene 100000                  BASELINE config 2
# Summaries land in gpurun_out/profiles_r6/ (copy into profiles/ and commit).
cd $GRAFT_REPO_ROOT
TAG=$1; SKIP=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out
P=$OUT/profiles_r6
mkdir -p $P
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --value-only $*"
cd /tmp && export TMPDIR=/tmp
D=/tmp/prof_r6_$TAG
rm -rf $D*
timeout 600 rocprofv3 --kernel-trace --stats -d ${D} -o t -- $B > $OUT/prof_r6_${TAG}_trace.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d ${D}_fetch -o f -- $B > $OUT/prof_r6_${TAG}_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d ${D}_write -o w -- $B > $OUT/prof_r6_${TAG}_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY -d ${D}_sq -o m -- $B > $OUT/prof_r6_${TAG}_sq.log 2>&1
timeout 900 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d ${D}_sq2 -o m -- $B > $OUT/prof_r6_${TAG}_sq2.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d ${D}_mix -o m -- $B > $OUT/prof_r6_${TAG}_mix.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $D/t_results.db $P/r6_${TAG}_kernel_trace_stats.txt > /dev/null
python tools/rocpd_series.py $D/t_results.db nn_stream4 50 > $P/r6_${TAG}_nn_in_loop_duration_series.txt
python tools/rocpd_summary.py ${D}_fetch/f_results.db $P/r6_${TAG}_pmc_FETCH_SIZE.txt > /dev/null
python tools/rocpd_summary.py ${D}_write/w_results.db $P/r6_${TAG}_pmc_WRITE_SIZE.txt > /dev/null
python tools/rocpd_summary.py ${D}_sq/m_results.db $P/r6_${TAG}_pmc_SQ.txt > /dev/null
python tools/make_pmc_json.py $D/t_results.db ${D}_fetch/f_results.db ${D}_write/w_results.db ${D}_sq/m_results.db $P/r6_${TAG}_pmc_summary.json $SKIP ${D}_sq2/m_results.db ${D}_mix/m_results.db > /dev/null
python - $P/r6_${TAG}_pmc_summary.json "$B" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); d["command"] = sys.argv[2]; json.dump(d, open(sys.argv[1], "w"), indent=1)
PY
head -20 $P/r6_${TAG}_kernel_trace_stats.txt; cat $P/r6_${TAG}_nn_in_loop_duration_series.txt
tail -2 $OUT/prof_r6_${TAG}_trace.log
