#!/bin/bash
set -x
cd /root/repo; mkdir -p gpurun_out; O=/root/repo/gpurun_out
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_issue_probe tools/valu_issue_probe.hip 2>/dev/null
/tmp/valu_issue_probe > $O/valu_issue_probe2.txt
cut -c1-60 $O/valu_issue_probe2.txt; awk '{print $1, $(NF-9), $(NF-8)}' $O/valu_issue_probe2.txt | head -0
python - <<'PY'
import re
for l in open('/root/repo/gpurun_out/valu_issue_probe2.txt'):
    m=re.findall(r'W=8 ([0-9.]+)',l)
    if m: print(l.split()[0], 'W=8', m[0])
PY
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_logging
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_logging -o r -- python3 /root/repo/bench.py --steps 2 --warmup 1 --value-only --profile-logging > $O/prof_logging.log 2>&1
python3 /root/repo/tools/rocpd_summary.py /tmp/prof_logging/r_results.db $O/r4_kernel_trace_logging.txt > /dev/null
head -14 $O/r4_kernel_trace_logging.txt
