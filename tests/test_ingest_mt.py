"""Multi-threaded ingest: plain FASTA / PAF files are mapped, cut at record boundaries and parsed by several
threads. Every array must equal what the streaming single-thread readers produce (which follow Contig.cpp:43-117,
Longread.cpp:109-162 and :234-302), and so must the error behaviour."""
import gzip
import shutil

import numpy as np
import pytest

from haslr_amd import host


def arrays(ds):
    c, r, h = ds.contigs, ds.reads, ds.hits
    out = {}
    n = h.n
    for name in ("q_id", "q_start", "q_end", "t_id", "t_len", "t_start", "t_end", "n_match", "n_block", "is_rev", "mapq"):
        out[name] = np.ctypeslib.as_array(getattr(h, name), shape=(n,)).copy() if n else np.zeros(0)
    out["cg_off"] = np.ctypeslib.as_array(h.cg_off, shape=(n + 1,)).copy()
    out["cg_ops"] = np.ctypeslib.as_array(h.cg_ops, shape=(int(out["cg_off"][-1]),)).copy() if out["cg_off"][-1] else np.zeros(0)
    out["read_len"] = np.ctypeslib.as_array(r.len, shape=(r.n,)).copy()
    out["read_off"] = np.ctypeslib.as_array(r.off, shape=(r.n + 1,)).copy()
    out["read_packed"] = np.ctypeslib.as_array(r.packed, shape=(int(out["read_off"][-1]),)).copy()
    out["read_hit_off"] = np.ctypeslib.as_array(ds.read_hit_off, shape=(r.n + 1,)).copy()
    out["contig_len"] = np.ctypeslib.as_array(c.len, shape=(c.n,)).copy()
    out["total"] = np.array([ds.total_read_bases])
    return out


def same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("threads", [2, 3, 8, 16])
def test_arrays_do_not_depend_on_thread_count(sim, threads):
    pre = sim("--genome-len", "250000", "--seed", "41")
    one = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf", threads=1)
    many = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf", threads=threads)
    same(arrays(one), arrays(many))
    one.close(); many.close()


def test_odd_text_is_read_the_same(sim, tmp_path):
    """CRLF line ends, blank lines, blanks inside sequence lines, wrapped sequences, no final newline, lower case, N"""
    pre = sim("--genome-len", "120000", "--seed", "42")
    reads = open(pre + ".reads.fa").read().split("\n")
    out = []
    for i, ln in enumerate(reads):
        if ln.startswith(">"):
            out.append(ln + (" some comment" if i % 3 == 0 else ""))
        elif ln:
            k = i % 5
            if k == 0:
                out.extend(ln[j:j + 70] for j in range(0, len(ln), 70))          # wrapped
            elif k == 1:
                out.append(ln[:len(ln) // 2] + " \t" + ln[len(ln) // 2:])         # blanks inside
            elif k == 2:
                out.append(ln.lower().replace("a", "n", 3))                       # lower case and N
            else:
                out.append(ln)
            if i % 7 == 0:
                out.append("")                                                    # blank line
    rf = tmp_path / "reads.fa"
    rf.write_bytes("\r\n".join(out).encode())                                     # CRLF, no final newline
    paf = open(pre + ".paf").read().rstrip("\n").replace("\n", "\r\n", 50)         # some CRLF lines, no final newline
    pf = tmp_path / "map.paf"
    pf.write_bytes((paf.replace("\r\n", "\r\n\r\n", 3)).encode())                 # and a few blank lines
    one = host.Dataset(pre + ".contigs.fa", str(rf), str(pf), threads=1)
    many = host.Dataset(pre + ".contigs.fa", str(rf), str(pf), threads=7)
    same(arrays(one), arrays(many))
    one.close(); many.close()


def test_gzip_and_fastq_fall_back_to_streaming(sim, tmp_path):
    pre = sim("--genome-len", "120000", "--seed", "43")
    gz = tmp_path / "reads.fa.gz"
    with open(pre + ".reads.fa", "rb") as fi, gzip.open(gz, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    fq = tmp_path / "reads.fq"
    with open(pre + ".reads.fa") as fi, open(fq, "w") as fo:
        name = None
        for ln in fi:
            ln = ln.rstrip("\n")
            if ln.startswith(">"):
                name = ln[1:]
            elif ln:
                fo.write("@%s\n%s\n+\n%s\n" % (name, ln, "@" * len(ln)))           # '@' quality: must not start records
    ref = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf", threads=1)
    for path in (gz, fq):
        ds = host.Dataset(pre + ".contigs.fa", str(path), pre + ".paf", threads=8)
        same(arrays(ref), arrays(ds))
        ds.close()
    ref.close()


@pytest.mark.parametrize("kind", ["columns", "number", "order", "query", "target"])
def test_errors_are_the_streaming_readers_errors(sim, tmp_path, kind):
    pre = sim("--genome-len", "250000", "--seed", "44")
    lines = open(pre + ".paf").read().rstrip("\n").split("\n")
    k = (len(lines) * 2) // 3                       # a line in the third quarter: another thread's part
    f = lines[k].split("\t")
    if kind == "columns":
        lines[k] = "\t".join(f[:9])
    elif kind == "number":
        f[7] = "12x"; lines[k] = "\t".join(f)
    elif kind == "order":
        f[0] = "0"; lines[k] = "\t".join(f)
    elif kind == "query":
        f[0] = "99999999"; lines[k] = "\t".join(f)
    else:
        f[5] = "99999999"; lines[k] = "\t".join(f)
    pf = tmp_path / "bad.paf"
    pf.write_text("\n".join(lines) + "\n")
    msgs = []
    for t in (1, 8):
        with pytest.raises(host.HostError) as e:
            host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", str(pf), threads=t)
        msgs.append(str(e.value))
    assert msgs[0] == msgs[1] and "[ERROR]" in msgs[0]
