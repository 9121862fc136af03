# in-loop NN launch time (mean over a fit, HIP events) + step time for library variants on ONE box: tools/ab_lib.sh "" _noslow ...
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  [ "$v" = "-" ] && v=""
  FDCAP_LIB=$GRAFT_REPO_ROOT/4dcapture-fpv_amd/libfdcap_hip$v.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-logging-run --no-exact-fp32 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('variant [$v]: %.2f ms/step  in-loop NN mean %.4f ms  steady %.4f ms' % (d['ms_per_step'], r['ms_per_launch'], r['steady_state_ms_per_launch']))"
done
