#!/usr/bin/env python3
"""Registers / spills / LDS / occupancy of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage, device pass only).
   python tools/kernel_resources.py [filter-substring] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else ""
extra = [a for a in sys.argv[1:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans", "-Xclang",
       "-target-feature", "-Xclang", "-packed-fp32-ops", "-DFDC_BUILD_NO_PK_F32", "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only",
       "-c", "-o", "/dev/null", *extra, os.path.join(ROOT, "4dcapture-fpv_amd", "csrc", "fdcap.hip")]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::|fdc::", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>6s} {'LDS':>7s} {'occ':>4s}")
for k, r in sorted(rows.items()):
    if flt in k:
        print(f"{k[:70]:70s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('VGPRs Spill', 0):6d} {r.get('LDS Size [bytes/block]', 0):7d} {r.get('Occupancy [waves/SIMD]', 0):4d}")
