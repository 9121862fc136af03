#!/bin/bash
# A/B of two builds of the library on one box in one call: libfdcap_hip.so (the tree) against libfdcap_hip_base.so
cd /root/repo; mkdir -p gpurun_out; O=/root/repo/gpurun_out
for i in 1 2 3; do
python bench.py --steps 5 --warmup 2 --value-only 2>/dev/null | tail -1 > $O/b_ab_new_$i.json
FDCAP_LIB=$PWD/4dcapture-fpv_amd/libfdcap_hip_base.so python bench.py --steps 5 --warmup 2 --value-only 2>/dev/null | tail -1 > $O/b_ab_base_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_ab_*.json')):
    d=json.loads(open(f).read()); print(f, round(d['value']), round(d['ms_per_step'],2))
PY
cd /tmp; export TMPDIR=/tmp
for m in new base; do
  rm -rf /tmp/prof_$m
  L=/root/repo/4dcapture-fpv_amd/libfdcap_hip.so; [ $m = base ] && L=/root/repo/4dcapture-fpv_amd/libfdcap_hip_base.so
  FDCAP_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -o r -- python3 /root/repo/bench.py --steps 2 --warmup 1 --value-only > $O/prof_ab_$m.log 2>&1
  python3 /root/repo/tools/rocpd_summary.py /tmp/prof_$m/r_results.db $O/r4_kernel_trace_ab_$m.txt > /dev/null
  echo "== $m"; sed -n 4,14p $O/r4_kernel_trace_ab_$m.txt | cut -c1-110
done
