"""tools/hxident (the size-independent check of tools/full_size_check.py and of two parity tests): a contig is placed on the strand that
most of its seeds vote for, not on the one its first seed happens to hit"""
import os
import random
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def test_strand_by_seed_vote(built, tmp_path):
    rng = random.Random(5)
    g = "".join(rng.choice("ACGT") for _ in range(300000))
    # an inverted repeat: the reverse complement of genome[200000:200400] also sits at the start of the region the first contig covers,
    # so the contig's first seeds hit the genome on BOTH strands
    g = g[:1000] + rc(g[200000:200400]) + g[1400:]
    (tmp_path / "g.fa").write_text(">g\n" + g + "\n")
    contigs = {"fwd_with_inverted_repeat": g[1000:90000], "reverse": rc(g[100000:190000]),
               "with_errors": "".join(c if rng.random() > 0.002 else rng.choice("ACGT") for c in rc(g[210000:290000])),
               "foreign": "".join(rng.choice("ACGT") for _ in range(3000))}
    (tmp_path / "a.fa").write_text("".join(">%s\n%s\n" % kv for kv in contigs.items()))
    out = subprocess.check_output([os.path.join(ROOT, "tools", "hxident"), str(tmp_path / "g.fa"), str(tmp_path / "a.fa")], text=True)
    rows = {ln.split("\t")[0]: ln.split("\t") for ln in out.split("\n") if "\t" in ln}
    assert rows["fwd_with_inverted_repeat"][2] == "+" and float(rows["fwd_with_inverted_repeat"][5].split("=")[1]) > 0.9999
    assert rows["reverse"][2] == "-" and float(rows["reverse"][5].split("=")[1]) > 0.9999
    assert rows["with_errors"][2] == "-" and 0.997 < float(rows["with_errors"][5].split("=")[1]) < 0.9995
    assert rows["foreign"][2] == "UNPLACED"
    summary = out.strip().split("\n")[-1].split()
    assert summary[0] == "identity" and float(summary[1]) > 0.998 and summary[-1] == "1"
