# plain + logging step for environment settings on ONE box: tools/ab_env5.sh "VAR=val" "-" ...
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  a=$(env $envs python bench.py --value-only --steps ${STEPS:-4} --warmup 1 $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])")
  b=$(env $envs python bench.py --value-only --profile-logging --steps ${STEPS:-4} --warmup 1 $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])")
  echo "[$cfg]: plain $a ms/step   logging $b ms/step"
done
