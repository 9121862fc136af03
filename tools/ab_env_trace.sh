# kernel-trace averages for environment settings on ONE box: tools/ab_env_trace.sh <kernel pattern> "VAR=val" "-" ...
cd /tmp && export TMPDIR=/tmp
K=$1; shift
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  rm -rf /tmp/abt
  env $envs timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/abt -o t -- python3 $GRAFT_REPO_ROOT/bench.py --value-only --steps 2 --warmup 1 $BENCH_ARGS > /tmp/abt.log 2>&1
  echo "[$cfg] $(tail -1 /tmp/abt.log | python3 -c "import json,sys; print('%.2f ms/step' % json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)"
  python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/abt/t_results.db /dev/null | grep -i "$K" | cut -c1-110
done
