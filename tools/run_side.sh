#!/bin/bash
set -x
cd /root/repo; mkdir -p gpurun_out; O=/root/repo/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "side_stream or deferred or ragged or trajectory or reproducible or checkpoint" > $O/t_side.log 2>&1; echo "tests rc=$?"; tail -15 $O/t_side.log
for i in 1 2; do
python bench.py --steps 3 --warmup 1 --no-exact-fp32 --no-cpu-baseline 2>/dev/null | tail -1 > $O/b_side_on_$i.json
FDCAP_LOG_OVERLAP=0 python bench.py --steps 3 --warmup 1 --no-exact-fp32 --no-cpu-baseline 2>/dev/null | tail -1 > $O/b_side_off_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_side_*.json')):
    d=json.loads(open(f).read()); print(f, round(d['value']), round(d['ms_per_step'],2), round(d['with_reference_logging']['value']), round(d['with_reference_logging']['ms_per_step'],2))
PY
