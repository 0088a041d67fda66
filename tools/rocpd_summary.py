#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / min / max (ns).
usage: rocpd_summary.py results.db > profiles/<name>_kernel_stats.txt"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
try:
    rows = list(c.execute("select * from top_kernels"))
    cols = [d[0] for d in c.execute("select * from top_kernels").description]
    print("# view top_kernels:", cols)
    for r in rows:
        print("\t".join(str(x) for x in r))
except Exception as e:  # noqa: BLE001
    print("# top_kernels unavailable:", e)
print()
q = """select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),
       max(d.workgroup_size_x), max(d.grid_size_x)
       from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""
tot = 0
rows = list(c.execute(q))
for r in rows:
    tot += r[2]
print("kernel\tcalls\ttotal_ns\tavg_ns\tmin_ns\tmax_ns\tpct\tworkgroup\tgrid")
for r in rows:
    print(f"{r[0][:90]}\t{r[1]}\t{r[2]}\t{r[3]:.0f}\t{r[4]}\t{r[5]}\t{100.0 * r[2] / tot:.2f}\t{r[6]}\t{r[7]}")
