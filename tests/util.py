"""Shared helpers for the test-suite (formatting the reference's dump formats from result arrays)."""
import hashlib
import os

import numpy as np


def sha256_bytes(b):
    return hashlib.sha256(b).hexdigest()


def sha256_file(p):
    h = hashlib.sha256()
    with open(p, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def dataset_arrays(ds):
    """numpy views of the raw-hit arrays of a haslr_amd.host.Dataset."""
    n = ds.hits.n
    g = lambda p, dt, m=n: np.ctypeslib.as_array(p, shape=(int(m),)).astype(dt, copy=False) if m else np.zeros(0, dt)
    h = {k: g(getattr(ds.hits, k), np.uint32) for k in ("q_id", "q_start", "q_end", "t_id", "t_len", "t_start", "t_end", "n_match", "n_block")}
    h["is_rev"] = g(ds.hits.is_rev, np.uint8)
    h["mapq"] = g(ds.hits.mapq, np.uint8)
    h["cg_off"] = g(ds.hits.cg_off, np.uint64, n + 1)
    h["cg_ops"] = g(ds.hits.cg_ops, np.uint32, int(h["cg_off"][-1]) if n else 0)
    return h


def cigar_text(ops, b, e, skf, skb):
    out = []
    for k in range(int(b), int(e)):
        w = int(ops[k])
        ln = w >> 2
        if k == b:
            ln -= int(skf)
        if k + 1 == e:
            ln -= int(skb)
        if ln > 0:
            c = "MID?"[w & 3]
            if out and out[-1][1] == c:          # the reference re-collapses a trimmed CIGAR (Common.cpp:123-150)
                out[-1] = (out[-1][0] + ln, c)
            else:
                out.append((ln, c))
    return "".join(f"{n}{c}" for n, c in out)


def alignments_paf(ds, chain):
    """Text of the reference's print_loaded_alignments (Longread.cpp:705-718) after fix_alignments."""
    h = dataset_arrays(ds)
    lines = []
    for a in range(len(chain["hit"])):
        x = int(chain["hit"][a])
        cg = cigar_text(h["cg_ops"], chain["cg_begin"][a], chain["cg_end"][a], chain["cg_skip_front"][a], chain["cg_skip_back"][a])
        lines.append("%u\t%u\t%u\t%c\t%u\t%u\t%u\t%u\t%u\t%u\tcg:Z:%s\n" % (
            h["q_id"][x], chain["q_start"][a], chain["q_end"][a], "-" if h["is_rev"][x] else "+", h["t_id"][x], chain["t_start"][a],
            chain["t_end"][a], chain["n_match"][a], chain["n_block"][a], h["mapq"][x], cg))
    return "".join(lines)


def edge_supp_text(edges, keep=None):
    """Text of oracle/ref_front_driver.cpp's dump_edge_supp for an edges_out dict; `keep` = set of edge keys."""
    lines = []
    ek, eo = edges["edge_key"], edges["edge_off"]
    for i in range(len(ek)):
        key = int(ek[i])
        if keep is not None and key not in keep:
            continue
        v, to = key >> 32, key & 0xffffffff
        b, e = int(eo[i]), int(eo[i + 1])
        parts = ["E\t%u\t%d\t%u\t%u\t%u" % (v >> 1, v & 1, to >> 1, to & 1, e - b)]
        for r in range(b, e):
            lr = int(edges["lr"][r])
            parts.append("%u:%u:%u:%u" % (lr & 0x7fffffff, lr >> 31, edges["cmp_head"][r], edges["cmp_tail"][r]))
        lines.append("\t".join(parts) + "\n")
    return "".join(lines)


def gfa_skeleton_text(path):
    out = []
    with open(path) as f:
        for line in f:
            if line.startswith("S\t"):
                p = line.rstrip("\n").split("\t")
                p[2] = str(len(p[2]))
                out.append("\t".join(p) + "\n")
            else:
                out.append(line)
    return "".join(out)


def gfa_edge_keys(path):
    """directed edge keys present in a GFA written by the pipeline"""
    keys = set()
    with open(path) as f:
        for line in f:
            if line.startswith("L\t"):
                p = line.split("\t")
                v = (int(p[1]) << 1) | (1 if p[2] == "-" else 0)
                to = (int(p[3]) << 1) | (1 if p[4] == "-" else 0)
                keys.add((v << 32) | to)
    return keys


def compare_dirs(a, b, names=None):
    """byte-compare files present in both dirs; returns list of differing names"""
    bad = []
    for f in sorted(os.listdir(a)):
        if names is not None and f not in names:
            continue
        pa, pb = os.path.join(a, f), os.path.join(b, f)
        if os.path.exists(pb) and open(pa, "rb").read() != open(pb, "rb").read():
            bad.append(f)
    return bad


class DeviceBallast:
    """device memory held for the duration of a test (hipMalloc through libamdhip64): what ELSE is resident on a GPU beside the thing under test"""

    def __init__(self, device=0):
        import ctypes as C
        self._C = C
        self._hip = C.CDLL("libamdhip64.so")
        self._hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self._hip.hipFree.argtypes = [C.c_void_p]
        self._hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        assert self._hip.hipSetDevice(device) == 0
        self._blocks = []
        self.bytes = 0

    def free_bytes(self):
        f, t = self._C.c_size_t(), self._C.c_size_t()
        assert self._hip.hipMemGetInfo(self._C.byref(f), self._C.byref(t)) == 0
        return int(f.value)

    def hold(self, n_bytes):
        """n_bytes more, in blocks of at most 8 GB"""
        left = int(n_bytes)
        while left > 0:
            n = min(left, 8 << 30)
            p = self._C.c_void_p()
            rc = self._hip.hipMalloc(self._C.byref(p), n)
            assert rc == 0, f"hipMalloc of {n} ballast bytes failed ({rc}); {self.free_bytes()} free"
            self._blocks.append(p)
            self.bytes += n
            left -= n

    def release(self):
        for p in self._blocks:
            self._hip.hipFree(p)
        self._blocks, self.bytes = [], 0


# BASELINE configs[3]'s data set (140 Mb, PacBio-like 25x, seed 0x4841534c + 3: bench.py --workload fly) as the session's simulator cache keys it - ONE argument
# list for every test that uses it, so that it is made once per session (test_gpu_parity.py checks it against the oracle, test_group_sharded.py runs the binary
# with eight ranks over it and removes the files)
CONFIGS3_ARGS = ("--genome-len", "140000000", "--seed", hex(0x4841534C + 3), "--model", "pacbio", "--cov", "25", "--variant-per-mb", "1.5")
