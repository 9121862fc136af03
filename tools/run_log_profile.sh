#!/bin/bash
# per-kernel cost of the every-iteration-logging step against the plain one (r4)
set -x
cd /root/repo; mkdir -p gpurun_out; O=/root/repo/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q > $O/t_log.log 2>&1; echo "tests rc=$?"
tail -4 $O/t_log.log
cd /tmp; export TMPDIR=/tmp
for m in plain logging; do
  F=""; [ $m = logging ] && F="--profile-logging"
  rm -rf /tmp/prof_$m
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -o r -- python3 /root/repo/bench.py --steps 2 --warmup 1 --value-only $F > $O/prof_$m.log 2>&1
  python3 /root/repo/tools/rocpd_summary.py /tmp/prof_$m/r_results.db $O/r4_kernel_trace_$m.txt > /dev/null || ls -R /tmp/prof_$m | head
done
head -16 $O/r4_kernel_trace_plain.txt; head -16 $O/r4_kernel_trace_logging.txt
cd /root/repo
for i in 1 2; do python bench.py --steps 3 --warmup 1 --no-exact-fp32 --no-cpu-baseline 2>/dev/null | tail -1 > $O/b_log_$i.json; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_log_*.json')):
    d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'], d['with_reference_logging']['value'], d['with_reference_logging']['ms_per_step'])
PY
