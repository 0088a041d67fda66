"""development (round 5): reads the `[hx-edge]` lines of an HX_DEBUG=2 run (tools/dev_r05.sh edgedump) and prints, per option set of the run, the
resident waves over time, the wave-slot time per workgroup shape, the last edges to end, and per shape the cycles per DP row of every phase.
Usage: dev_edge_timeline.py edgedump.err [number of option sets in the file]"""
import collections
import sys


def parse(path):
    out = []
    for line in open(path):
        if not line.startswith("[hx-edge]"):
            continue
        t = line.split()
        d = {t[i]: float(t[i + 1]) for i in range(2, len(t) - 1, 2)}
        d["e"] = int(t[1])
        out.append(d)
    return out


def hw(h):
    h = int(h)
    return (h >> 15) & 15, (h >> 13) & 3, (h >> 8) & 15   # XCC, SE, CU


PH = ("decode", "dp", "tb", "graph", "order", "csr")


def report(E, name):
    end = max(d["end_us"] for d in E)
    print(f"== {name}: {len(E)} edges, last end {end / 1e3:.1f} ms")
    ev = []
    for d in E:
        w = d["lanes"] / 64 * max(1, d["members"])
        ev += [(d["begin_us"], w), (d["end_us"], -w)]
    ev.sort()
    cur, last, acc = 0.0, 0.0, collections.defaultdict(float)
    for t, w in ev:
        a = last
        while a < t:
            b = min(t, (int(a // 50000) + 1) * 50000)
            acc[int(a // 50000)] += cur * (b - a)
            a = b
        cur += w
        last = t
    print("   resident waves, 50 ms bins:", " ".join(f"{acc[k] / 50000:.0f}" for k in sorted(acc)))
    if "hw" in E[0]:
        for T in (0.25 * end, 0.6 * end):
            cu = collections.Counter()
            for d in E:
                if d["begin_us"] <= T < d["end_us"]:
                    cu[hw(d["hw"])] += int(d["lanes"]) // 64 * max(1, int(d["members"]))
            hist = collections.Counter(cu.values())
            print(f"   waves per CU at {T / 1e3:.0f} ms ({len(cu)} CUs):", " ".join(f"{k}:{v}" for k, v in sorted(hist.items())))
    G = collections.defaultdict(collections.Counter)
    for d in E:
        k = (int(d["lanes"]), min(int(d["passes"]), 5) if d["members"] == 1 else -1)
        w = d["lanes"] / 64 * max(1, d["members"])
        g = G[k]
        for p in PH:
            g[p] += d[p] * w
        g["slot"] += (d["end_us"] - d["begin_us"]) * w
        g["rows"] += d["rows"]
        g["n"] += 1
        g["us"] += d["end_us"] - d["begin_us"]
        g["w"] += w * d["rows"]
        g["maxdur"] = max(g["maxdur"], d["end_us"] - d["begin_us"])
    tot = collections.Counter()
    print("   lanes, windows (5 = five or more, -1 = shared): edges, wave-slot seconds, longest chain, us per DP row | cycles per DP row and wave of every phase")
    for k in sorted(G):
        g = G[k]
        print(f"   {k[0]:5d} {k[1]:2d}: {g['n']:5d} edges {g['slot'] / 1e6:7.1f} slot-s, longest {g['maxdur'] / 1e3:6.1f} ms, {g['us'] / max(1, g['rows']):.2f} us/row | "
              + " ".join(f"{p} {g[p] / max(1, g['w']):.0f}" for p in PH[1:]))
        for p in PH:
            tot[p] += g[p]
    s = sum(tot.values())
    print(f"   all: {sum(g['slot'] for g in G.values()) / 1e6:.0f} wave-slot seconds; shares of the wave cycles: " + " ".join(f"{p} {100 * tot[p] / s:.1f} %" for p in PH))
    for d in sorted(E, key=lambda d: -d["end_us"])[:4]:
        print(f"   late: edge {d['e']} lmax {d['lmax']:.0f} nseq {d['nseq']:.0f} lanes {d['lanes']:.0f} windows {d['passes']:.0f} begin {d['begin_us'] / 1e3:.0f} end {d['end_us'] / 1e3:.0f} ms")


if __name__ == "__main__":
    E = parse(sys.argv[1])
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n = len(E) // ns
    for i in range(ns):
        report(E[i * n:(i + 1) * n], sys.argv[3 + i] if len(sys.argv) > 3 + i else f"option set {i}")
