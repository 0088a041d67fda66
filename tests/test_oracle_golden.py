"""The CPU oracle (and the product's host-side graph cleaning / writers) against golden vectors that were
produced by the UNMODIFIED reference front half (tests/golden/make_golden.py, oracle/_ref/ref_front).
This is what pins the oracle for SURVEY.md rows a1-a7 and a10."""
import gzip
import json
import os

import pytest

import orclib
import util
from haslr_amd import host

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(d for d in os.listdir(GOLD) if os.path.isfile(os.path.join(GOLD, d, "manifest.json")))


def run_oracle(pre_contigs, pre_reads, pre_paf, out):
    ds = host.Dataset(pre_contigs, pre_reads, pre_paf)
    be = orclib.OracleBackend(ds, 2)
    run = host.Run(ds, ds.params(), be.table, out)
    run.chain()
    run.graph()
    return ds, be, run


def check_against(case_dir, ds, run, out):
    man = json.load(open(os.path.join(case_dir, "manifest.json")))
    exp = os.path.join(case_dir, "expected")
    # files the pipeline writes itself: compact_uniq.txt, .stat, .log, GFA
    for f in sorted(os.listdir(exp)):
        if f.endswith(".gz") or f == "uniq_freq.txt":
            continue
        if f.endswith(".skel"):
            got = util.gfa_skeleton_text(os.path.join(out, f[:-5]))
            assert got == open(os.path.join(exp, f)).read(), f"{f[:-5]}: links / segment table differ from the reference"
            assert util.sha256_file(os.path.join(out, f[:-5])) == man["outputs"][f[:-5]], f"{f[:-5]}: bytes differ from the reference"
        else:
            assert open(os.path.join(out, f)).read() == open(os.path.join(exp, f)).read(), f"{f} differs from the reference"
    assert float(open(os.path.join(exp, "uniq_freq.txt")).read()) == ds.uniq_freq
    # internal states: overlap-trimmed alignments and per-edge support vectors
    chain, edges = run.chain_out(), run.edges_out(sides=False)
    assert util.sha256_bytes(util.alignments_paf(ds, chain).encode()) == man["outputs"]["alignments.fixed.paf"], "fix_alignments result differs"
    for tag, gfa in (("01", "backbone.01.init.gfa"), ("02", "backbone.02.weakEdge.gfa"), ("06", "backbone.06.smallbubble.gfa")):
        keep = None if tag == "01" else util.gfa_edge_keys(os.path.join(out, gfa))
        want = gzip.open(os.path.join(exp, f"edge_supp.{tag}.txt.gz"), "rt").read()
        assert util.edge_supp_text(edges, keep) == want, f"edge_supp.{tag} (support order) differs from the reference"


@pytest.mark.parametrize("case", [c for c in CASES if not c.startswith("committed_inputs")])
def test_generated_inputs_match_reference(case, sim, tmp_path):
    cd = os.path.join(GOLD, case)
    man = json.load(open(os.path.join(cd, "manifest.json")))
    pre = sim(*man["hxsim_args"])
    for k, h in man["inputs"].items():
        if util.sha256_file(pre + k) != h:
            pytest.skip("tools/hxsim produced different bytes than when the fixture was made (different libm?)")
    ds, be, run = run_oracle(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf", str(tmp_path / "o"))
    check_against(cd, ds, run, str(tmp_path / "o"))


@pytest.mark.parametrize("case", [c for c in CASES if c.startswith("committed_inputs")])
def test_committed_inputs_match_reference(case, built, tmp_path):
    cd = os.path.join(GOLD, case)
    reads = tmp_path / "reads.fa"
    with open(reads, "w") as f:   # the front half only needs read lengths
        for i, line in enumerate(open(os.path.join(cd, "inputs", "read_len.txt"))):
            f.write(f">{i}\n{'A' * int(line)}\n")
    ds, be, run = run_oracle(os.path.join(cd, "inputs", "in.contigs.fa.gz"), str(reads), os.path.join(cd, "inputs", "in.paf.gz"), str(tmp_path / "o"))
    check_against(cd, ds, run, str(tmp_path / "o"))
