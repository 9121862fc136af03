#!/bin/bash
set -x
cd /root/repo; mkdir -p gpurun_out; O=/root/repo/gpurun_out
python tools/phase2_motion.py > $O/phase2_motion.txt 2>&1; tail -9 $O/phase2_motion.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_logging
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_logging -o r -- python3 /root/repo/bench.py --steps 1 --warmup 0 --value-only --profile-logging > $O/prof_logging.log 2>&1
python3 /root/repo/tools/rocpd_series.py /tmp/prof_logging/r_results.db nn_stream4 25
