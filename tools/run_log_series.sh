#!/bin/bash
# the Chamfer search's duration launch by launch through ONE every-iteration-logging fit (400 phase-1 + 100 phase-2 searches)
cd /root/repo; mkdir -p gpurun_out; O=/root/repo/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_ls
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_ls -o r -- python3 /root/repo/bench.py --steps 1 --warmup 0 --value-only --profile-logging > $O/prof_ls.log 2>&1
python3 - <<'PY' > /root/repo/gpurun_out/nn_logging_series.txt 2>&1
import sqlite3, glob
c = sqlite3.connect(glob.glob('/tmp/prof_ls/*results.db')[0])
t = [r[0] for r in c.execute("select name from sqlite_master where name like 'rocpd_kernel_dispatch%'")][0]
sy = [r[0] for r in c.execute("select name from sqlite_master where name like 'rocpd_info_kernel_symbol%'")][0]
d = [r[0] / 1e3 for r in c.execute(f"select d.end-d.start from {t} d join {sy} s on d.kernel_id = s.id where s.kernel_name like '%nn_stream4%' order by d.start")]
print(f"nn_stream4 in one logging fit: {len(d)} launches, mean {sum(d)/len(d):.1f} us")
print("per block of 25:", " ".join(f"{sum(d[i:i+25])/len(d[i:i+25]):.1f}" for i in range(0, len(d), 25)))
print("launches 400..:", " ".join(f"{x:.0f}" for x in d[400:]))
rows = c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {t} d join {sy} s on d.kernel_id = s.id order by d.start").fetchall()
nn = [i for i, r in enumerate(rows) if 'nn_stream4' in r[0]]
print("dispatch sequence from the 399th to the 404th search (name, duration us, gap before it us, grid, workgroup):")
for i in range(nn[398], nn[404] + 1):
    n, a, b, g, wg = rows[i]
    print(f"  {n.split('(')[0][-44:]:44s} {(b - a) / 1e3:8.1f} {(a - rows[i - 1][2]) / 1e3:7.1f} {g:9d} {wg:5d}")
PY
cat /root/repo/gpurun_out/nn_logging_series.txt
