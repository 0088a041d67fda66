#!/bin/bash
# collects the per-version profile set on the GPU box: kernel-trace stats, the two PMC traffic passes, an unprofiled bench line
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_$c.log 2>&1
done
python - <<PY > $O/traffic.json
import sqlite3, glob, json
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob("$O/pmc_%s/**/*.db" % c, recursive=True)[0]
    con = sqlite3.connect(db); cur = con.cursor()
    T = {t[0].split('_0000')[0]: t[0] for t in cur.execute("select name from sqlite_master where type='table'")}
    q = f"select e.value, s.kernel_name from {T['rocpd_pmc_event']} e join {T['rocpd_kernel_dispatch']} d on e.event_id = d.event_id join {T['rocpd_info_kernel_symbol']} s on d.kernel_id = s.id join {T['rocpd_info_pmc']} i on e.pmc_id = i.id where i.name = '{c}'"
    res[c] = sum(v for v, k in cur.execute(q) if 'k_poa' in k)
out = {"workload": "bench.py default (E. coli-size), one step, k_poa dispatches summed", "FETCH_SIZE_raw_kb": res["FETCH_SIZE"], "WRITE_SIZE_raw_kb": res["WRITE_SIZE"],
       "hbm_bytes_raw": (res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024,
       "note": "separate --pmc passes with --kernel-trace only; counter unit KB (x1024); gfx950 caveat of MI355X_MICROARCH.md: FETCH_SIZE may under-count 128B requests"}
print(json.dumps(out, indent=1))
PY
timeout 600 python $R/bench.py > $O/bench.json 2> $O/bench.log
tail -3 $O/bench.log
cat $O/traffic.json
head -20 $O/kernel_stats.txt
