# A/B of environment knobs on ONE box: tools/ab_env.sh "VAR=val VAR2=val" "..." ...   ("-" = defaults)
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  r=$(env $envs python bench.py --value-only --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms/step  %.0f frames/s' % (d['ms_per_step'], d['value']))")
  echo "[$cfg] $r"
done
