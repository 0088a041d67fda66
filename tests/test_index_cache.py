"""index.contig / index.longread, pinned BOTH ways by the compiled reference (oracle/_ref/ref_front, whose harness can
load / write the caches with the reference's own read_*_index / write_*_index):
  product writes  -> the reference loads them and reproduces, byte for byte, what it derives from the text files
  reference writes -> the product loads them into the arrays it would filter out of the text files itself"""
import os
import subprocess

import numpy as np
import pytest

import orclib
from haslr_amd import host

FRONT_FILES = ["alignments.loaded.paf", "alignments.fixed.paf", "compact_uniq.txt", "edge_supp.01.txt", "edge_supp.06.txt", "uniq_freq.txt",
               "backbone.01.init.gfa", "backbone.06.smallbubble.gfa", "backbone.06.smallbubble.stat", "seqs.dump.txt"]


def run_ref(ref_front, pre, out, env_extra):
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, REF_FRONT_DUMP_SEQS="1", **env_extra)
    subprocess.check_call([ref_front, "-c", pre + ".contigs.fa", "-l", pre + ".reads.fa", "-m", pre + ".paf", "-d", out, "-t", "1"],
                          env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def product_indexes(pre, out_dir):
    """the product's writers on the chain output of the CPU oracle backend (the GPU path writes through the same host code)"""
    os.makedirs(out_dir, exist_ok=True)
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    be = orclib.OracleBackend(ds, 4)
    run = host.Run(ds, ds.params(), be.table, None)
    run.chain()
    ds.write_contig_index(os.path.join(out_dir, "index.contig"))
    run.write_longread_index(os.path.join(out_dir, "index.longread"))
    return ds, be, run


@pytest.mark.parametrize("seed,model", [("71", "pacbio"), ("72", "nanopore")])
def test_reference_loads_the_products_indexes(sim, ref_front, tmp_path, seed, model):
    pre = sim("--genome-len", "150000", "--seed", seed, "--model", model)
    a, b, idx = str(tmp_path / "text"), str(tmp_path / "fromidx"), str(tmp_path / "idx")
    run_ref(ref_front, pre, a, {})
    ds, be, run = product_indexes(pre, idx)
    run_ref(ref_front, pre, b, {"REF_FRONT_FROM_INDEX": idx})
    for f in FRONT_FILES:
        assert open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read(), f
    run.close(); be.close(); ds.close()


def survivors(ds, run):
    """(field arrays) of the alignments that survive the chain stage's filters, raw fields, read order"""
    ch = run.chain_out()
    hit = ch["hit"]
    h = ds.hits
    out = {}
    for name in ("q_id", "q_start", "q_end", "t_id", "t_len", "t_start", "t_end", "n_match", "n_block", "is_rev", "mapq"):
        out[name] = np.ctypeslib.as_array(getattr(h, name), shape=(h.n,))[hit].copy()
    cg_off = np.ctypeslib.as_array(h.cg_off, shape=(h.n + 1,))
    cg_ops = np.ctypeslib.as_array(h.cg_ops, shape=(max(1, int(cg_off[-1])),))
    out["cg"] = [tuple(cg_ops[cg_off[x]:cg_off[x + 1]]) for x in hit]
    out["read_off"] = ch["read_off"].copy()
    return out


@pytest.mark.parametrize("seed", ["73", "74"])
def test_product_loads_the_references_indexes(sim, ref_front, tmp_path, seed):
    pre = sim("--genome-len", "150000", "--seed", seed)
    refdir = str(tmp_path / "ref")
    run_ref(ref_front, pre, refdir, {"REF_FRONT_WRITE_INDEX": "1"})
    # text parse + own filters
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    be = orclib.OracleBackend(ds, 4)
    run = host.Run(ds, ds.params(), be.table, None)
    run.chain()
    want = survivors(ds, run)
    # the reference's caches
    di = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf", index_dir=refdir)
    assert di.used_contig_index and di.used_longread_index
    h = di.hits
    assert h.n == len(want["q_id"])
    for name in ("q_id", "q_start", "q_end", "t_id", "t_len", "t_start", "t_end", "n_match", "n_block", "is_rev", "mapq"):
        assert np.array_equal(np.ctypeslib.as_array(getattr(h, name), shape=(h.n,)), want[name]), name
    cg_off = np.ctypeslib.as_array(h.cg_off, shape=(h.n + 1,))
    cg_ops = np.ctypeslib.as_array(h.cg_ops, shape=(max(1, int(cg_off[-1])),))
    assert [tuple(cg_ops[cg_off[x]:cg_off[x + 1]]) for x in range(h.n)] == want["cg"]
    assert np.array_equal(np.ctypeslib.as_array(di.read_hit_off, shape=(di.reads.n + 1,)), want["read_off"])
    # sequences, lengths, k-mer statistics
    for a, b, n in ((ds.reads, di.reads, ds.reads.n), (ds.contigs, di.contigs, ds.contigs.n)):
        assert a.n == b.n
        assert np.array_equal(np.ctypeslib.as_array(a.len, shape=(n,)), np.ctypeslib.as_array(b.len, shape=(n,)))
    ro = np.ctypeslib.as_array(ds.reads.off, shape=(ds.reads.n + 1,))
    assert np.array_equal(ro, np.ctypeslib.as_array(di.reads.off, shape=(di.reads.n + 1,)))
    assert np.array_equal(np.ctypeslib.as_array(ds.reads.packed, shape=(int(ro[-1]),)), np.ctypeslib.as_array(di.reads.packed, shape=(int(ro[-1]),)))
    assert np.array_equal(np.ctypeslib.as_array(ds.contigs.mean_kmer, shape=(ds.contigs.n,)), np.ctypeslib.as_array(di.contigs.mean_kmer, shape=(di.contigs.n,)))
    assert ds.uniq_freq == di.uniq_freq and ds.total_read_bases == di.total_read_bases
    # and the whole pipeline gives the same assembly from either
    run.graph(); run.coords(); run.consensus(); run.assemble()
    be2 = orclib.OracleBackend(di, 4)
    r2 = host.Run(di, di.params(), be2.table, None)
    r2.all()
    assert run.assembly_fasta() == r2.assembly_fasta() and run.cns_out() == r2.cns_out()
    r2.close(); be2.close(); di.close(); run.close(); be.close(); ds.close()


def test_odd_cigar_text_is_kept(sim, ref_front, tmp_path):
    """cg:Z: strings that op words cannot spell (=/X letters, a zero-length op, a leading zero) reach index.longread as they were"""
    pre = sim("--genome-len", "120000", "--seed", "75")
    lines = open(pre + ".paf").read().rstrip("\n").split("\n")
    n = 0
    for i, ln in enumerate(lines):
        f = ln.split("\t")
        for k in range(12, len(f)):
            if f[k].startswith("cg:Z:") and "M" in f[k] and n < 40:
                cg = f[k][5:]
                j = cg.index("M")
                if n % 3 == 0:
                    cg = cg[:j] + "=" + cg[j + 1:]            # '=' instead of the first 'M': another letter, contig-only for the walks
                elif n % 3 == 1:
                    cg = "0I" + cg                            # zero-length op
                else:
                    cg = "0" + cg                             # leading zero
                f[k] = "cg:Z:" + cg
                n += 1
        lines[i] = "\t".join(f)
    paf = tmp_path / "odd.paf"
    paf.write_text("\n".join(lines) + "\n")
    a, b, idx = str(tmp_path / "text"), str(tmp_path / "fromidx"), str(tmp_path / "idx")
    os.makedirs(a); os.makedirs(idx)
    env = dict(os.environ, REF_FRONT_DUMP_SEQS="1")
    args = [ref_front, "-c", pre + ".contigs.fa", "-l", pre + ".reads.fa", "-m", str(paf), "-t", "1"]
    subprocess.check_call(args + ["-d", a], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", str(paf), threads=4)
    be = orclib.OracleBackend(ds, 4)
    run = host.Run(ds, ds.params(), be.table, None)
    run.chain()
    ds.write_contig_index(os.path.join(idx, "index.contig"))
    run.write_longread_index(os.path.join(idx, "index.longread"))
    os.makedirs(b)
    subprocess.check_call(args + ["-d", b], env=dict(env, REF_FRONT_FROM_INDEX=idx), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in ("alignments.loaded.paf", "alignments.fixed.paf", "compact_uniq.txt", "edge_supp.06.txt"):
        assert open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read(), f
    run.close(); be.close(); ds.close()


def test_rerun_from_the_index_is_idempotent(sim, ref_front, tmp_path):
    """reads cut down to ONE alignment by the palindrome rule (missed-adapter reads) keep it in index.longread; a rerun that loads the index
    must not put the records through the filters again (the '<= 1 hit' rule would drop them): compact_uniq.txt, the trimmed alignments and
    the assembly of the rerun equal those of the text run and the reference's own rerun from the same cache"""
    pre = sim("--genome-len", "150000", "--seed", "19", "--variant-per-mb", "20", "--cov", "14", "--hairpin-frac", "0.15")
    a, b = str(tmp_path / "text"), str(tmp_path / "rerun")
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    be = orclib.OracleBackend(ds, 4)
    run = host.Run(ds, ds.params(), be.table, a)
    run.all()
    ch = run.chain_out()
    n_single = int(np.sum(np.diff(ch["read_off"]) == 1))
    assert n_single > 0, "the fixture must hold reads that the palindrome rule cuts to one alignment"
    ds.write_contig_index(os.path.join(a, "index.contig"))
    run.write_longread_index(os.path.join(a, "index.longread"))
    ds2 = host.Dataset("/nonexistent/c.fa", "/nonexistent/r.fa", "/nonexistent/m.paf", index_dir=a)
    assert ds2.used_contig_index and ds2.used_longread_index
    be2 = orclib.OracleBackend(ds2, 4)
    run2 = host.Run(ds2, ds2.params(), be2.table, b)
    run2.all()
    ch2 = run2.chain_out()
    for k in ("q_start", "q_end", "t_start", "t_end", "n_match", "n_block", "read_off", "cmp_off", "cmp_aln"):
        assert np.array_equal(ch[k], ch2[k]), k
    for f in ("compact_uniq.txt", "asm.final.fa", "backbone.06.smallbubble.gfa"):
        assert open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read(), f
    # the reference, rerun from the same cache, writes the same compact_uniq.txt
    r = str(tmp_path / "ref")
    run_ref(ref_front, pre, r, {"REF_FRONT_FROM_INDEX": a})
    assert open(os.path.join(r, "compact_uniq.txt"), "rb").read() == open(os.path.join(a, "compact_uniq.txt"), "rb").read()
    for x in (run, run2, be, be2, ds, ds2):
        x.close()
