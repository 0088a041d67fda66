#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / min / max (ns).
usage: rocpd_summary.py results.db > profiles/<name>_kernel_stats.txt"""
import glob
import json
import os
import re
import sqlite3
import sys


def pmc_summary(dirs):
    """rocpd_summary.py --pmc DIR...  : one rocprofv3 --pmc pass per directory (counters collected in separate passes, kernel-trace only).
    Sums every counter over the k_poa dispatches of the run (all dimensions / XCDs) and prints derived shares to stderr, JSON to stdout:
      traffic: FETCH_SIZE / WRITE_SIZE are in KB (x 1024); gfx950 caveat (MI355X_MICROARCH.md): FETCH_SIZE counts wide coalesced reads at half
      SQ:      SQ_* cycle counters are quad-cycles summed over waves; shares are fractions of SQ_WAVE_CYCLES (WAIT_ANY = parked on s_waitcnt /
               barrier, WAIT_INST_ANY = issue stalls, ACTIVE_INST_* = issuing); VALU wave-instructions x 64 lanes / DP cells = lane-ops per cell"""
    tot, per_kernel, launches = {}, {}, {}
    for d in dirs:
        for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
            con = sqlite3.connect(db)
            cur = con.cursor()
            T = {t[0].split("_0000")[0]: t[0] for t in cur.execute("select name from sqlite_master where type='table'")}
            if "rocpd_pmc_event" not in T:
                continue
            q = (f"select i.name, s.kernel_name, d.workgroup_size_x, d.grid_size_x, sum(e.value), count(distinct d.dispatch_id) from {T['rocpd_pmc_event']} e "
                 f"join {T['rocpd_kernel_dispatch']} d on e.event_id = d.event_id join {T['rocpd_info_kernel_symbol']} s on d.kernel_id = s.id "
                 f"join {T['rocpd_info_pmc']} i on e.pmc_id = i.id group by i.name, s.kernel_name, d.workgroup_size_x, d.grid_size_x")
            for name, kern, wg, grid, val, n in cur.execute(q):
                if "k_poa" not in kern:
                    continue
                m = re.search(r"k_poaILi(\d+)ELi(\d+)ELb([01])", kern)
                short = f"k_poa<{m.group(1)},{m.group(2)}> {wg} lanes x {grid // max(1, wg)} workgroups" if m else kern[:40]
                tot[name] = tot.get(name, 0) + val
                per_kernel.setdefault(short, {})[name] = per_kernel.setdefault(short, {}).get(name, 0) + val
                launches[short] = n
    out = {"counters_summed_over_k_poa_dispatches": tot, "per_launch_class": per_kernel, "dispatches_per_class": launches}
    wc = tot.get("SQ_WAVE_CYCLES")
    if wc:
        sh = {k: tot[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS") if k in tot}
        out["share_of_wave_cycles"] = sh
        print("share of SQ_WAVE_CYCLES (k_poa):", {k: round(v, 3) for k, v in sh.items()}, file=sys.stderr)
        for kname, c in sorted(per_kernel.items()):
            if c.get("SQ_WAVE_CYCLES"):
                print(f"  {kname}: waves {c.get('SQ_WAVES', 0):.0f}, parked {c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:.2f}, issue-stalled {c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES']:.2f}, "
                      f"VALU-issuing {c.get('SQ_ACTIVE_INST_VALU', 0) / c['SQ_WAVE_CYCLES']:.2f}; VALU insts {c.get('SQ_INSTS_VALU', 0):.3g}, SALU {c.get('SQ_INSTS_SALU', 0):.3g}, LDS {c.get('SQ_INSTS_LDS', 0):.3g}", file=sys.stderr)
    if "SQ_BUSY_CYCLES" in tot and "GRBM_GUI_ACTIVE" in tot:
        out["note_busy"] = "SQ_BUSY_CYCLES is summed over shader engines / XCDs"
    if "FETCH_SIZE" in tot or "WRITE_SIZE" in tot:
        out["hbm_bytes_raw"] = (tot.get("FETCH_SIZE", 0) + tot.get("WRITE_SIZE", 0)) * 1024
        out["FETCH_SIZE_raw_kb"], out["WRITE_SIZE_raw_kb"] = tot.get("FETCH_SIZE"), tot.get("WRITE_SIZE")
        print(f"HBM traffic of the k_poa dispatches: read {tot.get('FETCH_SIZE', 0) * 1024 / 1e9:.2f} GB (raw; up to 2x after the gfx950 correction), written {tot.get('WRITE_SIZE', 0) * 1024 / 1e9:.2f} GB", file=sys.stderr)
    print(json.dumps(out, indent=1))


if len(sys.argv) > 1 and sys.argv[1] == "--pmc":
    pmc_summary(sys.argv[2:])
    sys.exit(0)

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
