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
    "nanopore_80k_s5": ["--genome-len", "80000", "--seed", "5", "--variant-per-mb", "30", "--cov", "14", "--model", "nanopore"],
    "committed_inputs_60k_s26": ["--genome-len", "60000", "--seed", "26", "--variant-per-mb", "40", "--cov", "14"],
}
SMALL = ("compact_uniq.txt", "uniq_freq.txt", "backbone.branching.log")


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
            man = {"hxsim_args": args, "inputs": {k: sha(pre + k) for k in (".contigs.fa", ".reads.fa", ".paf")}, "outputs": {}}
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
