#!/usr/bin/env python3
"""development: turn what tools/dev_profile_run.sh left under gpurun_out/<tag>/ into the files committed under profiles/ (bench.py reads the
*_traffic.json / *_sq_counters.json of its workload from there): tools/dev_profile_collect.py <tag>"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
LABEL = {"": ("yeast", "bench.py default (BASELINE configs[2], 12 Mb Nanopore-like 25x)"),
         "fly_": ("fly", "bench.py --workload fly on one GPU (BASELINE configs[3]'s data set, 140 Mb PacBio-like 25x: the many-edge regime)")}
for pfx, (wl, text) in LABEL.items():
    p = os.path.join(src, pfx + "pmc_summary.json")
    if os.path.exists(p):
        d = json.load(open(p))
        d["bench_workload"] = wl
        d["workload"] = text + ", one step per pass, k_poa dispatches summed; separate --pmc passes with --kernel-trace only"
        json.dump(d, open(os.path.join(dst, f"{tag}_{pfx}sq_counters.json"), "w"), indent=1)
        t = {"bench_workload": wl, "workload": d["workload"], "FETCH_SIZE_raw_kb": d.get("FETCH_SIZE_raw_kb"), "WRITE_SIZE_raw_kb": d.get("WRITE_SIZE_raw_kb"),
             "hbm_bytes_raw": d.get("hbm_bytes_raw"),
             "note": "separate --pmc passes with --kernel-trace only; counter unit KB (x1024); gfx950 caveat of MI355X_MICROARCH.md: FETCH_SIZE counts wide coalesced reads at half their bytes"}
        json.dump(t, open(os.path.join(dst, f"{tag}_{pfx}traffic.json"), "w"), indent=1)
    for name in ("kernel_stats.txt", "bench_under_rocprof.json"):
        q = os.path.join(src, pfx + name)
        if os.path.exists(q) and os.path.getsize(q):
            shutil.copy(q, os.path.join(dst, f"{tag}_{pfx}{name}"))
for name in ("bench.json", "gpu_tests_all.log", "gpu_tests.log"):
    q = os.path.join(src, name)
    if os.path.exists(q) and os.path.getsize(q):
        shutil.copy(q, os.path.join(dst, f"{tag}_{'gpu_tests_all.log' if name.startswith('gpu_tests') else name}"))
print(sorted(f for f in os.listdir(dst) if f.startswith(tag)))
