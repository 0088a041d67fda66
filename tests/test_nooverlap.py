"""minia_nooverlap (haslr_amd/csrc/host/nooverlap.cpp) against the reference's tool: byte-identical output for every case
of tests/nooverlaplib.py — from the committed outputs of the compiled reference tool (tests/golden/nooverlap/, made by
tests/golden/make_nooverlap_golden.py) and, when oracle/_ref/ref_nooverlap is there, from running it beside the product on more inputs."""
import gzip
import os
import subprocess

import pytest

import nooverlaplib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "haslr_amd", "bin", "minia_nooverlap")
REF = os.path.join(ROOT, "oracle", "_ref", "ref_nooverlap")
GOLDEN = os.path.join(ROOT, "tests", "golden", "nooverlap")


def run(tool, path, k):
    pr = subprocess.run([tool, path, k], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return pr.returncode, pr.stdout


@pytest.mark.parametrize("name", sorted(nooverlaplib.cases()))
def test_committed_reference_outputs(built, name):
    base = os.path.join(GOLDEN, name)
    k = open(base + ".k").read()
    # the committed input is what the generator made (so the fixtures cannot drift from nooverlaplib)
    assert open(base + ".in", newline="").read() == nooverlaplib.cases()[name][0]
    rc, out = run(TOOL, base + ".in", k)
    if open(base + ".rc").read() == "0":
        assert rc == 0 and out == open(base + ".out", "rb").read()
    else:
        assert rc != 0


@pytest.mark.skipif(not os.path.isfile(REF), reason="oracle/_ref/ref_nooverlap not built (needs /root/reference in the build container)")
@pytest.mark.parametrize("seed", range(12))
def test_beside_the_compiled_reference(built, tmp_path, seed):
    k = [21, 31, 49, 50, 63, 7][seed % 6]
    text = nooverlaplib.minia_like(100 + seed, n=300, k=k, crlf=seed % 4 == 1, wrap=[0, 60, 80][seed % 3], blank_lines=seed % 5 == 2, tabs=seed % 2 == 1)
    p = tmp_path / "asm.fa"
    p.write_text(text, newline="")
    want = run(REF, str(p), str(k))
    assert want[0] == 0 and run(TOOL, str(p), str(k)) == want
    gz = tmp_path / "asm.fa.gz"
    with gzip.open(gz, "wb") as f:
        f.write(text.encode())
    assert run(TOOL, str(gz), str(k)) == want and run(REF, str(gz), str(k)) == want


def test_usage_and_errors(built, tmp_path):
    for tool in [TOOL] + ([REF] if os.path.isfile(REF) else []):
        pr = subprocess.run([tool, "-h"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert pr.returncode == 0 and pr.stdout == b"" and pr.stderr == b"usage: ./nooverlap unitigs.fa kmerSize\n"
        pr = subprocess.run([tool, "only_one"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert pr.returncode == 1 and pr.stderr == b"usage: ./nooverlap unitigs.fa kmerSize\n"
        pr = subprocess.run([tool], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert pr.returncode == 1
        missing = str(tmp_path / "nope.fa")
        pr = subprocess.run([tool, missing, "31"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert pr.returncode == 1 and pr.stderr == ("[ERROR] could not open file: %s\n" % missing).encode()
