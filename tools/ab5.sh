# round 5: plain step AND every-iteration-logging step for library variants on ONE box:  tools/ab5.sh - _base _b1 ...   ("-" = the tree's build)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  [ "$v" = "-" ] && v=""
  L=$GRAFT_REPO_ROOT/4dcapture-fpv_amd/libfdcap_hip$v.so
  a=$(FDCAP_LIB=$L python bench.py --value-only --steps ${STEPS:-4} --warmup 1 $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])")
  b=$(FDCAP_LIB=$L python bench.py --value-only --profile-logging --steps ${STEPS:-4} --warmup 1 $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])")
  echo "variant [$v]: plain $a ms/step   logging $b ms/step"
done
