#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; O=/root/repo/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_parity.py -x -q -k "rccl or loop_inside or checkpoint or verbose" 2>&1 | tail -3
(python tools/host_issue_probe.py 128; FDCAP_FORCE_EXCHANGE=1 python tools/host_issue_probe.py 128; python tools/host_issue_probe.py 1024) 2>&1 | grep "^frames\|RCCL version\|Librccl" > $O/r4_host_issue.txt
cat $O/r4_host_issue.txt
