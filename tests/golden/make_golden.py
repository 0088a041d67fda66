#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ — run in the BUILD container only
(needs /root/reference through oracle/_ref/ref_front, built by `make -C oracle ref`).

For every case: inputs are produced by tools/hxsim (seeded, deterministic), the UNMODIFIED reference front
half (oracle/_ref/ref_front = the reference's own Longread/Backbone_graph/Cleaning code) is run on them, and
its outputs are stored as data:
  manifest.json          generator arguments, sha256 of the generated inputs and of every reference output
  expected/*.txt|log|stat  the reference's small outputs verbatim
  expected/*.gfa.skel    GFA with the sequence column replaced by its length (full-file sha256 in the manifest)
  expected/*.gz          larger dumps (edge_supp; alignments.fixed.paf when small) gzip-compressed; the rest by sha256
The `committed_inputs` case additionally stores contigs + PAF (+ read lengths) so that the front half can be
replayed without the generator.
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CASES = {
    "pacbio_100k_s12": ["--genome-len", "100000", "--seed", "12", "--variant-per-mb", "40", "--cov", "14"],
    # Nanopore-like with missed-adapter (hairpin) reads; every counter of RICH below must be non-zero for it (checked when the fixture is made)
    "nanopore_rich_300k_s5": ["--genome-len", "300000", "--seed", "5", "--variant-per-mb", "60", "--cov", "16", "--model", "nanopore", "--hairpin-frac", "0.06"],
    "committed_inputs_60k_s26": ["--genome-len", "60000", "--seed", "26", "--variant-per-mb", "40", "--cov", "14"],
}
SMALL = ("compact_uniq.txt", "uniq_freq.txt", "backbone.branching.log")
RICH = ("nanopore_rich_300k_s5",)   # cases that must exercise every rule of the front half


def counters(pre, rd):
    """how often each rule of the front half fired in the reference run `rd` on the inputs `pre.*`: overlap trims, tips, simple / super /
    small bubbles, weak edges, and reads cut by the palindrome rule (a unique contig hit twice on one read, Longread.cpp:182-232)"""
    loaded, fixed = open(os.path.join(rd, "alignments.loaded.paf")).read().split("\n"), open(os.path.join(rd, "alignments.fixed.paf")).read().split("\n")
    c = {"trimmed_alignments": sum(a != b for a, b in zip(loaded, fixed))}
    c["tips"] = sum(1 for _ in open(os.path.join(rd, "backbone.03.tip.log")))
    c["simple_bubbles"] = open(os.path.join(rd, "backbone.04.simplebubble.log")).read().count("simple_bubble")
    c["super_bubbles"] = open(os.path.join(rd, "backbone.05.superbubble.log")).read().count("bubble_src")
    c["small_bubbles"] = open(os.path.join(rd, "backbone.06.smallbubble.log")).read().count("small_bubble")
    n1 = sum(1 for ln in open(os.path.join(rd, "backbone.01.init.gfa")) if ln.startswith("L"))
    n2 = sum(1 for ln in open(os.path.join(rd, "backbone.02.weakEdge.gfa")) if ln.startswith("L"))
    c["weak_edges"] = (n1 - n2) // 2
    uniq = float(open(os.path.join(rd, "uniq_freq.txt")).read())
    km = {}
    for ln in open(pre + ".contigs.fa"):
        if ln.startswith(">"):
            f = ln[1:].split()
            km[f[0]] = float([x for x in f if x.startswith("km:f:")][0][5:])
    per_read = {}
    for ln in open(pre + ".paf"):
        f = ln.split("\t")
        if int(f[10]) < 500 or int(f[9]) / int(f[10]) < 0.85 or int(f[11]) < 55 or km[f[5]] > uniq * 3.15:
            continue
        per_read.setdefault(f[0], []).append(f[5])
    c["palindrome_reads"] = sum(1 for v in per_read.values() if any(v.count(t) > 1 and km[t] < uniq * 1.15 for t in set(v)))
    return c


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for b in iter(lambda: f.read(1 << 20), b""):
            h.update(b)
    return h.hexdigest()


def gfa_skeleton(src, dst):
    with open(src) as f, open(dst, "w") as o:
        for line in f:
            if line.startswith("S\t"):
                p = line.rstrip("\n").split("\t")
                p[2] = str(len(p[2]))
                o.write("\t".join(p) + "\n")
            else:
                o.write(line)


def main():
    sim, ref = os.path.join(ROOT, "tools", "hxsim"), os.path.join(ROOT, "oracle", "_ref", "ref_front")
    if not os.path.exists(ref):
        sys.exit("oracle/_ref/ref_front missing: run `make -C oracle ref` where /root/reference exists")
    for name, args in CASES.items():
        out = os.path.join(HERE, name)
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(os.path.join(out, "expected"))
        with tempfile.TemporaryDirectory() as d:
            pre = os.path.join(d, "in")
            subprocess.check_call([sim] + args + ["--out-prefix", pre], stderr=subprocess.DEVNULL)
            rd = os.path.join(d, "ref")
            os.makedirs(rd)
            subprocess.check_call([ref, "-c", pre + ".contigs.fa", "-l", pre + ".reads.fa", "-m", pre + ".paf", "-d", rd],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            man = {"hxsim_args": args, "inputs": {k: sha(pre + k) for k in (".contigs.fa", ".reads.fa", ".paf")}, "outputs": {}, "counters": counters(pre, rd)}
            if name in RICH:
                assert all(v > 0 for v in man["counters"].values()), (name, man["counters"])
            for f in sorted(os.listdir(rd)):
                p = os.path.join(rd, f)
                man["outputs"][f] = sha(p)
                if f.endswith(".gfa"):
                    gfa_skeleton(p, os.path.join(out, "expected", f + ".skel"))
                elif f.endswith((".stat", ".log")) or f in SMALL:
                    shutil.copy(p, os.path.join(out, "expected", f))
                elif f.startswith("edge_supp") or (f == "alignments.fixed.paf" and os.path.getsize(p) < 400000):
                    with open(p, "rb") as fi, gzip.GzipFile(os.path.join(out, "expected", f + ".gz"), "wb", mtime=0) as fo:
                        shutil.copyfileobj(fi, fo)
            if name.startswith("committed_inputs"):
                os.makedirs(os.path.join(out, "inputs"))
                for k in (".contigs.fa", ".paf"):
                    with open(pre + k, "rb") as fi, gzip.GzipFile(os.path.join(out, "inputs", "in" + k + ".gz"), "wb", mtime=0) as fo:
                        shutil.copyfileobj(fi, fo)
                with open(pre + ".reads.fa") as fi, open(os.path.join(out, "inputs", "read_len.txt"), "w") as fo:
                    for line in fi:
                        if not line.startswith(">"):
                            fo.write(f"{len(line.strip())}\n")
            with open(os.path.join(out, "manifest.json"), "w") as f:
                json.dump(man, f, indent=1, sort_keys=True)
        print(name, "ok")


if __name__ == "__main__":
    main()
