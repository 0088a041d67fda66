"""Oracle + host pipeline against the compiled reference front half on fresh seeded inputs (byte equality of
every file the reference writes before Assemble.cpp). oracle/_ref/ref_front travels with the repo snapshot."""
import os
import subprocess

import pytest

import orclib
import util
from haslr_amd import host

CASES = [("--genome-len", "70000", "--seed", "101", "--variant-per-mb", "40", "--cov", "12"),
         ("--genome-len", "90000", "--seed", "202", "--variant-per-mb", "30", "--cov", "10", "--model", "nanopore"),
         ("--genome-len", "60000", "--seed", "303", "--cov", "12", "--model", "perfect", "--variant-per-mb", "20")]


@pytest.mark.parametrize("args", CASES, ids=lambda a: "_".join(a[1:4:2]))
def test_front_half_files_identical(args, sim, ref_front, tmp_path):
    pre = sim(*args)
    rd, od = tmp_path / "ref", tmp_path / "orc"
    rd.mkdir()
    subprocess.check_call([ref_front, "-c", pre + ".contigs.fa", "-l", pre + ".reads.fa", "-m", pre + ".paf", "-d", str(rd)],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    be = orclib.OracleBackend(ds, 2)
    run = host.Run(ds, ds.params(), be.table, str(od))
    run.chain()
    run.graph()
    assert util.compare_dirs(str(rd), str(od)) == []
    assert len([f for f in os.listdir(od) if f.endswith(".gfa")]) == 6
    chain, edges = run.chain_out(), run.edges_out(sides=False)
    assert util.alignments_paf(ds, chain) == open(rd / "alignments.fixed.paf").read()
    assert util.edge_supp_text(edges) == open(rd / "edge_supp.01.txt").read()
    keep = util.gfa_edge_keys(str(od / "backbone.06.smallbubble.gfa"))
    assert util.edge_supp_text(edges, keep) == open(rd / "edge_supp.06.txt").read()


def test_nondefault_parameters(sim, ref_front, tmp_path):
    pre = sim("--genome-len", "70000", "--seed", "404", "--variant-per-mb", "40", "--cov", "12")
    rd, od = tmp_path / "ref", tmp_path / "orc"
    rd.mkdir()
    flags = ["--aln-block", "800", "--aln-sim", "0.87", "--edge-sup", "2", "--uniq-dev", "0.1"]
    subprocess.check_call([ref_front, "-c", pre + ".contigs.fa", "-l", pre + ".reads.fa", "-m", pre + ".paf", "-d", str(rd)] + flags,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    be = orclib.OracleBackend(ds, 1)
    run = host.Run(ds, ds.params(min_aln_block=800, min_aln_sim=0.87, min_edge_sup=2, max_uniq_dev=0.1), be.table, str(od))
    run.chain()
    run.graph()
    assert util.compare_dirs(str(rd), str(od)) == []


@pytest.mark.parametrize("threads", ["1", "3", "16"])
def test_parallel_cleaning_passes_equal_the_reference(threads, sim, ref_front, tmp_path):
    """the four cleaning passes with their per-node tests evaluated by several host threads and the edits committed in the
    reference's node order (host/graph.cpp) - every GFA, stat and log byte-equal to the compiled reference's (Cleaning.cpp:7-184, :488-648),
    on a graph where every pass fires (tips, simple / super / small bubbles, weak edges), whatever the thread count"""
    pre = sim("--genome-len", "1500000", "--seed", "5", "--variant-per-mb", "60", "--cov", "16", "--model", "nanopore", "--hairpin-frac", "0.06")
    rd, od = tmp_path / "ref", tmp_path / "orc"
    rd.mkdir()
    subprocess.check_call([ref_front, "-c", pre + ".contigs.fa", "-l", pre + ".reads.fa", "-m", pre + ".paf", "-d", str(rd)],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    be = orclib.OracleBackend(ds, 8)
    old = os.environ.get("HASLR_CLEAN_THREADS")
    try:
        os.environ["HASLR_CLEAN_THREADS"] = threads
        run = host.Run(ds, ds.params(), be.table, str(od))
        run.chain()
        run.graph()
    finally:
        if old is None:
            os.environ.pop("HASLR_CLEAN_THREADS", None)
        else:
            os.environ["HASLR_CLEAN_THREADS"] = old
    assert util.compare_dirs(str(rd), str(od)) == []
    logs = {f: open(od / f).read() for f in os.listdir(od) if f.endswith(".log")}
    assert logs["backbone.03.tip.log"].count("tip_len") >= 20 and "bubble_src" in logs["backbone.05.superbubble.log"]
    assert "simple_bubble" in logs["backbone.04.simplebubble.log"] and logs["backbone.06.smallbubble.log"].count("small_bubble") >= 10
    run.close(); be.close(); ds.close()
