#!/bin/bash
# A/B of two builds of the library on one box in one call: libfdcap_hip.so (the tree) against libfdcap_hip_base.so (build it from
# another commit: git worktree add /tmp/base_wt <commit>; hipcc ... -o 4dcapture-fpv_amd/libfdcap_hip_base.so /tmp/base_wt/.../fdcap.hip)
cd /root/repo; mkdir -p gpurun_out; O=/root/repo/gpurun_out
python tools/compare_builds.py 4dcapture-fpv_amd/libfdcap_hip.so 4dcapture-fpv_amd/libfdcap_hip_base.so 2>&1 | grep -v amdgpu.ids | tail -3
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ops_autograd.py tests/test_gpu_sharded.py -x -q 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
for m in new base; do
  rm -rf /tmp/prof_$m
  L=/root/repo/4dcapture-fpv_amd/libfdcap_hip.so; [ $m = base ] && L=/root/repo/4dcapture-fpv_amd/libfdcap_hip_base.so
  FDCAP_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -o r -- python3 /root/repo/bench.py --steps 2 --warmup 1 --value-only > $O/prof_ab_$m.log 2>&1
  python3 /root/repo/tools/rocpd_summary.py /tmp/prof_$m/r_results.db $O/r4_kernel_trace_ab_$m.txt > /dev/null
  echo "== $m"; sed -n 4,14p $O/r4_kernel_trace_ab_$m.txt | cut -c1-110
done
