"""Known-answer tests of the POA restatement (the reference has no tests at this boundary: SPOA 1.1.3 is
un-vendored; these are authored here) and end-to-end sanity of the unpinned rows a8/a9."""
import os
import random
import subprocess

import orclib
from haslr_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_identical_sequences():
    s = "ACGTTGCAAGGCTTAACCGGTACGATCGATTAGC"
    assert orclib.poa_consensus([s] * 5) == s
    assert orclib.poa_consensus([s]) == s


def test_empty_and_single_base():
    assert orclib.poa_consensus([]) == ""
    assert orclib.poa_consensus(["", ""]) == ""
    assert orclib.poa_consensus(["A"]) == "A"
    assert orclib.poa_consensus(["", "ACGT", ""]) == "ACGT"


def test_majority_substitution():
    a = "ACGTACGTACGTTTGACCAGTACGGATC"
    b = a[:10] + ("A" if a[10] != "A" else "C") + a[11:]
    assert orclib.poa_consensus([a, b, a]) == a
    assert orclib.poa_consensus([b, a, a]) == a
    assert orclib.poa_consensus([b, b, a]) == b


def test_majority_indel():
    a = "ACGTACGTACGTTTGACCAGTACGGATCAAGGCT"
    dele = a[:12] + a[15:]
    ins = a[:12] + "GGG" + a[12:]
    assert orclib.poa_consensus([a, dele, a, a]) == a
    assert orclib.poa_consensus([dele, dele, a, dele]) == dele
    assert orclib.poa_consensus([ins, a, a, a, ins]) == a
    assert orclib.poa_consensus([ins, ins, a, ins]) == ins


def test_noisy_reads_recover_template():
    rnd = random.Random(5)
    tmpl = "".join(rnd.choice("ACGT") for _ in range(400))

    def noisy():
        out = []
        for c in tmpl:
            r = rnd.random()
            if r < 0.03:
                continue
            if r < 0.05:
                out.append(rnd.choice("ACGT"))
            else:
                out.append(c)
            if rnd.random() < 0.05:
                out.append(rnd.choice("ACGT"))
        return "".join(out)

    cons = orclib.poa_consensus([noisy() for _ in range(25)])
    # edit distance
    prev = list(range(len(tmpl) + 1))
    for i, a in enumerate(cons, 1):
        cur = [i]
        for j, b in enumerate(tmpl, 1):
            cur.append(min(prev[j] + 1, cur[-1] + 1, prev[j - 1] + (a != b)))
        prev = cur
    assert prev[-1] <= 4, f"consensus is {prev[-1]} edits away from the template"


def test_error_free_reads_reassemble_the_genome(sim, tmp_path):
    """perfect reads: every consensus must be exact, so each assembled contig is a substring of the genome"""
    pre = sim("--genome-len", "120000", "--seed", "9", "--model", "perfect", "--no-variants", "--cov", "12")
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    be = orclib.OracleBackend(ds, 2)
    run = host.Run(ds, ds.params(), be.table, str(tmp_path / "o"))
    run.all()
    genome = open(pre + ".genome.fa").read().split("\n")[1]
    comp = str.maketrans("ACGT", "TGCA")
    recs = [l for l in run.assembly_fasta().split("\n") if l and not l.startswith(">")]
    assert recs and sum(map(len, recs)) > 0.6 * len(genome)
    for s in recs:
        assert s in genome or s.translate(comp)[::-1] in genome


def test_noisy_assembly_identity_vs_truth(sim, tmp_path):
    pre = sim("--genome-len", "120000", "--seed", "10", "--no-variants", "--cov", "25")
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    be = orclib.OracleBackend(ds, 4)
    run = host.Run(ds, ds.params(), be.table, str(tmp_path / "o"))
    run.all()
    out = subprocess.check_output([os.path.join(ROOT, "tools", "hxident"), pre + ".genome.fa", str(tmp_path / "o" / "asm.final.fa")], text=True)
    ident = float(out.strip().split("\n")[-1].split()[1])
    assert ident >= 0.998, out
