#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / min / max (ns).
usage: rocpd_summary.py results.db > profiles/<name>_kernel_stats.txt"""
import re
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

# the POA launch group: every dispatch on its own (the classes run concurrently on separate streams; bench.py's
# roofline.kernel_ms_per_launch is the HIP-event time around the whole group = its longest member, the per-name average above mixes classes)
print()
print("# k_poa dispatches in time order: start_ms (since the first)\tduration_ms\tworkgroup\tworkgroups\tkernel")
q2 = """select d.start, d.end - d.start, d.workgroup_size_x, d.grid_size_x, s.kernel_name
        from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id where s.kernel_name like '%k_poa%' order by d.start"""
rows2 = list(c.execute(q2))
t0 = rows2[0][0] if rows2 else 0
groups = []
for st, dur, wg, grid, name in rows2:
    m = re.search(r"k_poaILi(\d+)ELi(\d+)ELb([01])", name)
    short = f"k_poa<{m.group(1)},{m.group(2)},{'true' if m.group(3) == '1' else 'false'}>" if m else name[name.find("k_poa"):][:40]
    print(f"{(st - t0) / 1e6:.3f}\t{dur / 1e6:.3f}\t{wg}\t{grid // max(1, wg)}\t{short}")
    if not groups or st > groups[-1][1]:
        groups.append([st, st + dur])
    else:
        groups[-1][1] = max(groups[-1][1], st + dur)
if groups:
    print("# launch groups (overlapping dispatches merged): " + ", ".join(f"{(b - a) / 1e6:.1f} ms" for a, b in groups))
