#!/usr/bin/env python3
"""development: start / end of every k_poa launch of a rocprofv3 --kernel-trace database (which launch class ends when): tools/dev_trace_classes.py DIR_OR_DB"""
import glob, os, re, sqlite3, sys
p = sys.argv[1]
db = p if p.endswith(".db") else sorted(glob.glob(os.path.join(p, "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%k_poa%' order by d.start").fetchall()
t0 = None
for n, s, e, g, w in rows:
    m = re.search(r"k_poaILi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)", n) or re.search(r"k_poa<(\d+), (\d+), (\w+), (\w+), (\w+)>", n)
    nm = "x".join(m.groups()[:2]) + ("/persistent" if m.group(4) in ("1", "true") else "") + ("/pk16" if m.group(5) in ("1", "true") else "") if m else n[:40]
    if t0 is None or s - t0 > 3e8 and s > last_end:
        t0 = s; print("--- call")
    last_end = max(e, locals().get("last_end", 0))
    print(f"  {nm:24s} {w:5d} lanes x {g // w:5d} workgroups: {(s - t0) / 1e6:8.1f} -> {(e - t0) / 1e6:8.1f} ms")
