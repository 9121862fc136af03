# in-loop NN launch time (bench.py roofline.ms_per_launch) for the timing-ablation builds, with and without the work-list cache
cd $GRAFT_REPO_ROOT
for v in "" _st4a1 _st4a2; do
 for slack in 0.04 0; do
  FDCAP_LIB=$GRAFT_REPO_ROOT/4dcapture-fpv_amd/libfdcap_hip$v.so FDCAP_NN_CACHE_SLACK=$slack timeout 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-logging-run 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('variant \'$v\' slack $slack: in-loop NN %.4f ms' % d['roofline']['ms_per_launch'])"
 done
done
