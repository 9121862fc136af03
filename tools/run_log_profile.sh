#!/bin/bash
# per-kernel cost of the every-iteration-logging step against the plain one (r4)
set -x
cd /root/repo; mkdir -p gpurun_out; O=/root/repo/gpurun_out
cd /tmp; export TMPDIR=/tmp
for m in plain logging; do
  F=""; [ $m = logging ] && F="--profile-logging"
  rm -rf /tmp/prof_$m
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -o r -- python3 /root/repo/bench.py --steps 2 --warmup 1 --value-only $F > $O/prof_$m.log 2>&1
  python3 /root/repo/tools/rocpd_summary.py /tmp/prof_$m/r_results.db $O/r4_kernel_trace_$m.txt > /dev/null || ls -R /tmp/prof_$m | head
done
head -24 $O/r4_kernel_trace_plain.txt; head -24 $O/r4_kernel_trace_logging.txt; tail -2 $O/prof_plain.log $O/prof_logging.log
