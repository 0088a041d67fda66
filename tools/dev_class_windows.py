"""development (round 6): per pass of an HX_DEBUG=2 run (stderr of tools/dev_r05_ab.py), when the edges of every launch class began and ended"""
import sys, collections
passes, cur = [], []
for line in open(sys.argv[1]):
    if line.startswith("[hx-edge]"):
        t = line.split(); d = {t[i]: float(t[i + 1]) for i in range(2, len(t) - 1, 2)}; cur.append(d)
    elif ("] pass " in line or "cold first pass" in line or "] step " in line) and cur:
        passes.append((line.strip()[:60], cur)); cur = []
for name, E in passes:
    end = max(d["end_us"] for d in E)
    by = collections.defaultdict(list)
    for d in E: by[(int(d["cls"]), int(d["lanes"]))].append(d)
    print(name, "| last end %.0f ms" % (end / 1e3))
    if end < 560e3 and "--all" not in sys.argv: continue
    for k in sorted(by):
        v = by[k]; b = sorted(x["begin_us"] for x in v)
        print("   class %d lanes %4d: %5d edges, first begin %.1f ms, 10th pct %.1f, median %.1f, last begin %.1f, last end %.1f ms" % (k[0], k[1], len(v), b[0] / 1e3, b[len(b) // 10] / 1e3, b[len(b) // 2] / 1e3, b[-1] / 1e3, max(x["end_us"] for x in v) / 1e3))

    def hw(h):
        h = int(h); return "xcc%d se%d cu%d" % ((h >> 15) & 15, (h >> 13) & 3, (h >> 8) & 15)
    for k in sorted(by):
        if k[1] >= 512:
            for d in sorted(by[k], key=lambda x: x["begin_us"]):
                print("      edge lmax %d nseq %d passes %d begin %.1f ms end %.1f ms chain %.1f ms %s" % (d["lmax"], d["nseq"], d["passes"], d["begin_us"] / 1e3, d["end_us"] / 1e3, (d["end_us"] - d["begin_us"]) / 1e3, hw(d["hw"])))
    for d in sorted(E, key=lambda x: -x["end_us"])[:8]:
        print("      late edge: class %d lanes %d members %d lmax %d nseq %d passes %d begin %.1f ms end %.1f ms %s" % (d["cls"], d["lanes"], d.get("members", 0), d["lmax"], d["nseq"], d["passes"], d["begin_us"] / 1e3, d["end_us"] / 1e3, hw(d["hw"])))
