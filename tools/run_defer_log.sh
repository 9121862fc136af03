#!/bin/bash
# A/B of the like-for-like figure with the deferred step on logging iterations (r4)
set -x
cd /root/repo; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "deferred or reproducible or golden or trajectory" > gpurun_out/t_defer.log 2>&1; echo "defer rc=$?"
tail -5 gpurun_out/t_defer.log
for i in 1 2; do
python bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/b_defer_on_$i.json
FDCAP_DEFER_STEP=0 python bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/b_defer_off_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_defer_*.json')):
    d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'], d.get('with_reference_logging'))
PY
