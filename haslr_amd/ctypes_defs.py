"""ctypes mirrors of include/haslr_types.h and haslr_amd/csrc/host/haslr_host.h.

Field order and types must match the C headers exactly; tests/test_abi.py checks the sizes
against `sizeof` values exported by the libraries.
"""
import ctypes as C

import numpy as np

u8p, u32p, u64p, f64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_uint64, C.c_double))


class Params(C.Structure):
    _fields_ = [("min_aln_block", C.c_uint32), ("min_aln_sim", C.c_double), ("min_aln_mapq", C.c_uint32),
                ("max_uniq_dev", C.c_double), ("min_edge_sup", C.c_uint32), ("uniq_freq", C.c_double)]


class Contigs(C.Structure):
    _fields_ = [("n", C.c_uint32), ("mean_kmer", f64p), ("len", u32p)]


class Reads(C.Structure):
    _fields_ = [("n", C.c_uint32), ("len", u32p), ("off", u64p), ("packed", u8p)]


class Hits(C.Structure):
    _fields_ = [("n", C.c_uint64)] + [(k, u32p) for k in
                                      ("q_id", "q_start", "q_end", "t_id", "t_len", "t_start", "t_end", "n_match", "n_block")] + \
               [("is_rev", u8p), ("mapq", u8p), ("cg_off", u64p), ("cg_ops", u32p)]


class ChainOut(C.Structure):
    _fields_ = [("n_aln", C.c_uint64), ("n_reads", C.c_uint32), ("hit", u32p)] + \
               [(k, u32p) for k in ("q_start", "q_end", "t_start", "t_end", "n_match", "n_block")] + \
               [("cg_begin", u64p), ("cg_end", u64p), ("cg_skip_front", u32p), ("cg_skip_back", u32p),
                ("read_off", u64p), ("n_cmp", C.c_uint64), ("cmp_off", u64p), ("cmp_aln", u32p)]


class RecSide(C.Structure):
    _fields_ = [(k, u32p) for k in ("q_start", "q_end", "t_start", "t_end")] + \
               [("is_rev", u8p), ("cg_begin", u64p), ("cg_end", u64p), ("cg_skip_front", u32p), ("cg_skip_back", u32p)]


class EdgesOut(C.Structure):
    _fields_ = [("n_rec", C.c_uint64), ("key", u64p), ("lr", u32p), ("cmp_head", u32p), ("cmp_tail", u32p),
                ("head", RecSide), ("tail", RecSide), ("n_edge", C.c_uint64), ("edge_key", u64p), ("edge_off", u64p)]


class CoordsOut(C.Structure):
    _fields_ = [("n_edge", C.c_uint32), ("head_end", u32p), ("tail_beg", u32p), ("supp_off", u64p),
                ("supp_lr", u32p), ("spos", u32p), ("epos", u32p)]


class CnsOut(C.Structure):
    _fields_ = [("n_edge", C.c_uint32), ("cns_off", u64p), ("cns", C.POINTER(C.c_char)),
                ("dp_cells", C.c_uint64), ("seq_bases", C.c_uint64), ("n_aligned", C.c_uint64)]


class PoaParams(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap", C.c_int32)]


class Backend(C.Structure):
    _fields_ = [("ctx", C.c_void_p)] + [(k, C.c_void_p) for k in
                                        ("chain_reads", "edge_support", "edge_coords", "poa_batch", "free_chain",
                                         "free_edges", "free_coords", "free_cns", "last_error")]


def default_params(uniq_freq, min_aln_block=500, min_aln_sim=0.85, max_uniq_dev=0.15, min_edge_sup=3):
    """Defaults of Commandline.cpp:46-66 (MAPQ 55 is not exposed on the reference's command line)."""
    return Params(min_aln_block, min_aln_sim, 55, max_uniq_dev, min_edge_sup, uniq_freq)


def arr(ptr, n, dtype):
    """numpy copy of a C array (n elements)."""
    n = int(n)
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n,)).copy()


def chain_to_dict(c):
    n, nr = c.n_aln, c.n_reads
    d = {k: arr(getattr(c, k), n, np.uint32) for k in
         ("hit", "q_start", "q_end", "t_start", "t_end", "n_match", "n_block", "cg_skip_front", "cg_skip_back")}
    d.update({k: arr(getattr(c, k), n, np.uint64) for k in ("cg_begin", "cg_end")})
    d["read_off"] = arr(c.read_off, nr + 1, np.uint64)
    d["cmp_off"] = arr(c.cmp_off, nr + 1, np.uint64)
    d["cmp_aln"] = arr(c.cmp_aln, c.n_cmp, np.uint32)
    return d


def side_to_dict(s, n, prefix):
    d = {prefix + k: arr(getattr(s, k), n, np.uint32) for k in ("q_start", "q_end", "t_start", "t_end", "cg_skip_front", "cg_skip_back")}
    d[prefix + "is_rev"] = arr(s.is_rev, n, np.uint8)
    d[prefix + "cg_begin"] = arr(s.cg_begin, n, np.uint64)
    d[prefix + "cg_end"] = arr(s.cg_end, n, np.uint64)
    return d


def edges_to_dict(e, sides=True):
    n = e.n_rec
    d = {"key": arr(e.key, n, np.uint64), "lr": arr(e.lr, n, np.uint32), "cmp_head": arr(e.cmp_head, n, np.uint32),
         "cmp_tail": arr(e.cmp_tail, n, np.uint32), "edge_key": arr(e.edge_key, e.n_edge, np.uint64),
         "edge_off": arr(e.edge_off, e.n_edge + 1, np.uint64)}
    if sides:
        d.update(side_to_dict(e.head, n, "head_"))
        d.update(side_to_dict(e.tail, n, "tail_"))
    return d


def coords_to_dict(c):
    n = c.n_edge
    off = arr(c.supp_off, n + 1, np.uint64)
    m = int(off[-1]) if n + 1 else 0
    return {"head_end": arr(c.head_end, n, np.uint32), "tail_beg": arr(c.tail_beg, n, np.uint32), "supp_off": off,
            "supp_lr": arr(c.supp_lr, m, np.uint32), "spos": arr(c.spos, m, np.uint32), "epos": arr(c.epos, m, np.uint32)}


def cns_to_list(c):
    off = arr(c.cns_off, c.n_edge + 1, np.uint64)
    raw = C.string_at(c.cns, int(off[-1])) if c.n_edge else b""
    return [raw[int(off[i]):int(off[i + 1])].decode() for i in range(c.n_edge)]
