"""Regenerates tests/golden/nooverlap/<case>.in / .k / .out / .rc: the cases of tests/nooverlaplib.py through the REFERENCE's minia_nooverlap
compiled from its own source where it lies (oracle/_ref/ref_nooverlap, built by oracle/Makefile).
    python tests/golden/make_nooverlap_golden.py        (in the build container)"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import nooverlaplib  # noqa: E402

REF = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "ref_nooverlap")


def main():
    d = os.path.join(HERE, "nooverlap")
    os.makedirs(d, exist_ok=True)
    for name, (text, k) in nooverlaplib.cases().items():
        base = os.path.join(d, name)
        with open(base + ".in", "w", newline="") as f:
            f.write(text)
        open(base + ".k", "w").write(k)
        pr = subprocess.run([REF, base + ".in", k], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        open(base + ".out", "wb").write(pr.stdout)
        open(base + ".rc", "w").write("0" if pr.returncode == 0 else "nonzero")
        print(name, pr.returncode, len(pr.stdout))


if __name__ == "__main__":
    main()
