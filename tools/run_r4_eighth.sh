set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_repro tools/pk_f32_mfma_repro.hip 2>/dev/null && timeout 900 /tmp/pk_repro 40 > $O/pk_repro.txt 2>&1
tail -4 $O/pk_repro.txt
FDC_PK=+ bash tools/build_variant.sh pk > $O/build_pk.log 2>&1
timeout 900 python tools/pk_bisect.py 300 pk > $O/pk_bisect.txt 2>&1
cat $O/pk_bisect.txt | grep -v Warning | tail -3
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --value-only"
for f in 1 0; do
  rm -rf /tmp/prof_d$f
  FDCAP_DEFER_STEP=$f timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_d$f -o t -- $B > $O/prof_d$f.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof_d$f/t_results.db $O/defer${f}_kernels.txt > /dev/null
  head -16 $O/defer${f}_kernels.txt
done
