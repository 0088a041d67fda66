"""Python mirror of the host pipeline (libhaslr_host.so): ingest, graph cleaning, stitching, writers.

Stage names follow the reference's main() (main.cpp:115-219): chain -> graph -> coords -> consensus ->
assemble. The compute backend is a `ctypes_defs.Backend` table; the product fills it from
libhaslr_hip.so (`haslr_amd.hip.HipContext.backend()`).
"""
import ctypes as C
import os

from . import ctypes_defs as T

_LIBDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_LIBDIR, "libhaslr_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (or make -C haslr_amd/csrc)")
        L = C.CDLL(path)
        L.hxh_last_error.restype = C.c_char_p
        L.hxh_dataset_load.restype = C.c_void_p
        L.hxh_dataset_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        L.hxh_dataset_load_mt.restype = C.c_void_p
        L.hxh_dataset_load_mt.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_uint]
        L.hxh_dataset_load_cached.restype = C.c_void_p
        L.hxh_dataset_load_cached.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_uint, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.hxh_dataset_write_contig_index.argtypes = [C.c_void_p, C.c_char_p]
        L.hxh_run_write_longread_index.argtypes = [C.c_void_p, C.c_char_p]
        L.hxh_dataset_free.argtypes = [C.c_void_p]
        L.hxh_dataset_views.argtypes = [C.c_void_p, C.POINTER(T.Contigs), C.POINTER(T.Reads), C.POINTER(T.Hits), C.POINTER(T.u64p)]
        L.hxh_dataset_uniq_freq.restype = C.c_double
        L.hxh_dataset_uniq_freq.argtypes = [C.c_void_p]
        L.hxh_dataset_total_read_bases.restype = C.c_uint64
        L.hxh_dataset_total_read_bases.argtypes = [C.c_void_p]
        L.hxh_run_create.restype = C.c_void_p
        L.hxh_run_create.argtypes = [C.c_void_p, C.POINTER(T.Params), C.POINTER(T.Backend), C.c_char_p]
        L.hxh_run_free.argtypes = [C.c_void_p]
        for name in ("chain", "graph", "coords", "consensus", "assemble", "all"):
            getattr(L, "hxh_run_" + name).argtypes = [C.c_void_p]
        L.hxh_run_set_edge_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.hxh_run_set_read_shard.argtypes = [C.c_void_p, C.c_uint32]
        L.hxh_run_results_export.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64)]
        L.hxh_run_results_import.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        L.hxh_run_results_missing.argtypes = [C.c_void_p]
        L.hxh_run_results_missing.restype = C.c_uint64
        L.hxh_run_compact_text.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.hxh_run_compact_text.restype = C.POINTER(C.c_char)
        L.hxh_run_n_edges_total.argtypes = [C.c_void_p]
        L.hxh_run_n_edges_total.restype = C.c_uint32
        L.hxh_run_timings.argtypes = [C.c_void_p, C.POINTER(C.c_double * 5)]
        L.hxh_run_n_edges.argtypes = [C.c_void_p]
        L.hxh_run_n_edges.restype = C.c_uint32
        for name, ty in (("chain_out", T.ChainOut), ("edges_out", T.EdgesOut), ("coords_out", T.CoordsOut), ("cns_out", T.CnsOut)):
            f = getattr(L, "hxh_run_" + name)
            f.argtypes = [C.c_void_p]
            f.restype = C.POINTER(ty)
        L.hxh_shard_bounds.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.hxh_runs_all_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
        L.hxh_run_set_async_writers.argtypes = [C.c_void_p, C.c_int]
        L.hxh_run_assembly_fasta.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.hxh_run_assembly_fasta.restype = C.POINTER(C.c_char)
        _lib = L
    return _lib


class HostError(RuntimeError):
    pass


class Dataset:
    """Parsed inputs resident in host memory (contigs, packed long reads, raw PAF records)."""

    def __init__(self, contigs, reads, paf, long_fofn=False, mapping_fofn=False, threads=0, index_dir=None):
        """threads: ingest threads (0 = automatic, 1 = the streaming single-thread readers); the arrays do not depend on it.
        index_dir: a directory whose index.contig / index.longread (the reference's cache files) are loaded instead of the text files
        when they exist, like haslr_assemble does with its output directory."""
        L = lib()
        self.used_contig_index = self.used_longread_index = False
        if index_dir is None:
            self._h = L.hxh_dataset_load_mt(os.fsencode(contigs), os.fsencode(reads), int(long_fofn), os.fsencode(paf), int(mapping_fofn), int(threads))
        else:
            a, b = C.c_int(0), C.c_int(0)
            self._h = L.hxh_dataset_load_cached(os.fsencode(index_dir), os.fsencode(contigs), os.fsencode(reads), int(long_fofn), os.fsencode(paf),
                                                int(mapping_fofn), int(threads), C.byref(a), C.byref(b))
            self.used_contig_index, self.used_longread_index = bool(a.value), bool(b.value)
        if not self._h:
            raise HostError(L.hxh_last_error().decode())
        self.contigs, self.reads, self.hits = T.Contigs(), T.Reads(), T.Hits()
        self.read_hit_off = T.u64p()
        L.hxh_dataset_views(self._h, C.byref(self.contigs), C.byref(self.reads), C.byref(self.hits), C.byref(self.read_hit_off))
        self.uniq_freq = L.hxh_dataset_uniq_freq(self._h)
        self.total_read_bases = L.hxh_dataset_total_read_bases(self._h)

    def params(self, **kw):
        return T.default_params(self.uniq_freq, **kw)

    def write_contig_index(self, path):
        if lib().hxh_dataset_write_contig_index(self._h, os.fsencode(path)) != 0:
            raise HostError(lib().hxh_last_error().decode())

    def close(self):
        if self._h:
            lib().hxh_dataset_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_bounds(dataset, n):
    """n + 1 boundaries of contiguous read-id ranges with about equal numbers of raw PAF records (hxh_shard_bounds)"""
    b = (C.c_uint32 * (n + 1))()
    lib().hxh_shard_bounds(dataset._h, n, b)
    return list(b)


def runs_all_sharded(runs, read_begin):
    """The whole stage over `runs` inside this process, one host thread per rank (hxh_runs_all_sharded: what haslr_assemble --gpus N runs).
    runs[r] sits on rank r's backend table; runs[0] owns the output directory."""
    hs = (C.c_void_p * len(runs))(*[r._h for r in runs])
    rb = (C.c_uint32 * len(runs))(*read_begin[:len(runs)])
    if lib().hxh_runs_all_sharded(hs, len(runs), rb, None, None) != 0:
        raise HostError(lib().hxh_last_error().decode())


class Run:
    """One execution of the stage. `out_dir=None` writes no files (timing of the compute path)."""

    def __init__(self, dataset, params, backend, out_dir=None):
        self._ds, self._be, self._prm = dataset, backend, params   # keep alive
        if out_dir is not None:
            os.makedirs(out_dir, exist_ok=True)
        self._h = lib().hxh_run_create(dataset._h, C.byref(params), C.byref(backend), os.fsencode(out_dir) if out_dir else None)

    def _call(self, name):
        if getattr(lib(), "hxh_run_" + name)(self._h) != 0:
            raise HostError(lib().hxh_last_error().decode())

    def chain(self): self._call("chain")
    def graph(self): self._call("graph")
    def coords(self): self._call("coords")
    def consensus(self): self._call("consensus")
    def assemble(self): self._call("assemble")
    def all(self): self._call("all")

    def set_edge_shard(self, rank, world):
        lib().hxh_run_set_edge_shard(self._h, rank, world)

    def set_read_shard(self, lr_begin):
        lib().hxh_run_set_read_shard(self._h, lr_begin)

    def results_export(self):
        """bytes: coordinates + consensus of this run's share of the edges (multi-GPU)"""
        p, n = C.POINTER(C.c_uint8)(), C.c_uint64()
        if lib().hxh_run_results_export(self._h, C.byref(p), C.byref(n)) != 0:
            raise HostError(lib().hxh_last_error().decode())
        return C.string_at(p, n.value)

    def results_import(self, blob):
        if lib().hxh_run_results_import(self._h, blob, len(blob)) != 0:
            raise HostError(lib().hxh_last_error().decode())

    @property
    def results_missing(self): return lib().hxh_run_results_missing(self._h)

    def compact_text(self):
        n = C.c_uint64()
        p = lib().hxh_run_compact_text(self._h, C.byref(n))
        return C.string_at(p, n.value)

    @property
    def n_edges_total(self): return lib().hxh_run_n_edges_total(self._h)

    def write_longread_index(self, path):
        """index.longread (the reference's cache file): needs chain()"""
        if lib().hxh_run_write_longread_index(self._h, os.fsencode(path)) != 0:
            raise HostError(lib().hxh_last_error().decode())

    def timings(self):
        t = (C.c_double * 5)()
        lib().hxh_run_timings(self._h, C.byref(t))
        return dict(zip(("chain", "graph", "coords", "consensus", "assemble"), t))

    @property
    def n_edges(self): return lib().hxh_run_n_edges(self._h)
    def chain_out(self): return T.chain_to_dict(lib().hxh_run_chain_out(self._h).contents)
    def edges_out(self, sides=True): return T.edges_to_dict(lib().hxh_run_edges_out(self._h).contents, sides)
    def coords_out(self): return T.coords_to_dict(lib().hxh_run_coords_out(self._h).contents)
    def cns_out(self): return T.cns_to_list(lib().hxh_run_cns_out(self._h).contents)
    def cns_stats(self):
        c = lib().hxh_run_cns_out(self._h).contents
        return {"dp_cells": c.dp_cells, "seq_bases": c.seq_bases, "n_aligned": c.n_aligned}

    def assembly_fasta(self):
        n = C.c_uint64()
        p = lib().hxh_run_assembly_fasta(self._h, C.byref(n))
        return C.string_at(p, n.value).decode()

    def close(self):
        if self._h:
            lib().hxh_run_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
