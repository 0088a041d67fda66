"""development (round 6): the binary end to end on the 140 Mb data set, several runs, its own stage account (HASLR_STAGE_TIMES); HX_* options from the environment"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

wl = bench.WORKLOADS[os.environ.get("AB_WORKLOAD", "fly")]
pre = bench.make_dataset(wl, int(wl["genome"]), "gpu")
for spec in sys.argv[1:] or ["-"]:
    env = dict(kv.split("=") for kv in spec.split(",")) if spec != "-" else {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    r = bench.cli_e2e(pre, "fly")
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    st = r["stages_s"]
    print(f"[{spec}] e2e {r['cli_e2e_s']:.2f} s; " + ", ".join(f"{k[:-2]} {v:.2f}" for k, v in st.items() if v >= 0.05), flush=True)
