#!/bin/bash
# kernel trace of the inner fit's four optimiser variants (tools/innerfit_bench.py)
cd /root/repo; mkdir -p gpurun_out; O=/root/repo/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_if
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_if -o r -- python3 /root/repo/tools/innerfit_bench.py 64 > $O/prof_if.log 2>&1
tail -5 $O/prof_if.log
python3 - <<'PY' > /root/repo/gpurun_out/innerfit_kernels.txt 2>&1
import sqlite3, glob
db = glob.glob('/tmp/prof_if/*results.db')[0]
c = sqlite3.connect(db)
try:
    rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name order by 3 desc limit 14").fetchall()
except Exception:
    t = [r[0] for r in c.execute("select name from sqlite_master where name like 'rocpd_kernel_dispatch%'")][0]
    s = [r[0] for r in c.execute("select name from sqlite_master where name like 'rocpd_info_kernel_symbol%'")][0]
    rows = c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from {t} d join {s} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc limit 14").fetchall()
try:
    t = [r[0] for r in c.execute("select name from sqlite_master where name like 'rocpd_kernel_dispatch%'")][0]
    sy = [r[0] for r in c.execute("select name from sqlite_master where name like 'rocpd_info_kernel_symbol%'")][0]
    d = sorted(r[0] / 1e3 for r in c.execute(f"select d.end-d.start from {t} d join {sy} s on d.kernel_id = s.id where s.kernel_name like '%lbfgs_advance%'"))
    q = lambda f: d[min(len(d) - 1, int(f * len(d)))]
    print(f"# lbfgs_advance_kernel durations (us): min {d[0]:.1f}, 10 % {q(0.1):.1f}, median {q(0.5):.1f}, 90 % {q(0.9):.1f}, max {d[-1]:.1f}")
except Exception as e:
    print("# (no per-dispatch table:", e, ")")
print("# rocprofv3 --kernel-trace of tools/innerfit_bench.py 64 (all four variants, 3 repeats each): calls, total us, avg us")
for n, k, tot, avg in rows:
    print(f"{k:7d} {tot:12.1f} {avg:9.2f}  {n[:100]}")
PY
cat /root/repo/gpurun_out/innerfit_kernels.txt
