"""Python mirror of the C-ABI in include/haslr_hip.h (libhaslr_hip.so, gfx950 kernels).

No fallback: if the library or a HIP device is missing, construction raises.
"""
import ctypes as C
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # POA classes run on separate streams that must map to distinct hardware queues

from . import ctypes_defs as T

_LIBDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
_lib = None

SYMBOLS = ["hx_last_error", "hx_device_count", "hx_ctx_create", "hx_ctx_destroy", "hx_upload", "hx_set_read_shard", "hx_set_prefiltered",
           "hx_chain_reads", "hx_edge_support", "hx_edge_coords", "hx_poa_batch", "hx_free_chain", "hx_free_edges",
           "hx_free_coords", "hx_free_cns", "hx_edge_emit", "hx_edge_records_bytes", "hx_edge_records_export",
           "hx_edge_records_import", "hx_poa_supports", "hx_poa_sequences", "hx_timing_reset", "hx_timing_get", "hx_set_poa_block", "hx_backend_fill", "hx_poa_phase_cycles", "hx_set_poa_traceback", "hx_poa_workspace_bytes",
           "hx_set_option", "hx_get_option", "hx_option_names", "hx_poa_memory_stats", "hx_poa_release_workspace", "hx_poa_prune_stats", "hx_group_set_timeout", "hx_group_inject_fault", "hx_poa_reserve", "hx_poa_host_times", "hx_poa_arena_stats", "hx_group_rccl_ranks",
           "hx_group_create", "hx_group_destroy", "hx_group_size", "hx_group_ctx", "hx_group_transport", "hx_edge_merge", "hx_group_backend_fill", "hx_group_exchange_stats"]


class HipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_LIBDIR, "libhaslr_hip.so")
        if not os.path.exists(path):
            raise HipError(f"{path} is missing: the HIP extension must be built (python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        L = C.CDLL(path)
        L.hx_last_error.restype = C.c_char_p
        L.hx_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.hx_ctx_destroy.argtypes = [C.c_void_p]
        L.hx_upload.argtypes = [C.c_void_p, C.POINTER(T.Contigs), C.POINTER(T.Reads), C.POINTER(T.Hits), T.u64p]
        L.hx_set_read_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.hx_set_prefiltered.argtypes = [C.c_void_p, C.c_int]
        L.hx_chain_reads.argtypes = [C.c_void_p, C.POINTER(T.Params), C.POINTER(T.ChainOut)]
        L.hx_edge_support.argtypes = [C.c_void_p, C.POINTER(T.Params), C.POINTER(T.EdgesOut)]
        L.hx_edge_coords.argtypes = [C.c_void_p, C.c_uint32, T.u32p, C.POINTER(T.CoordsOut)]
        L.hx_poa_batch.argtypes = [C.c_void_p, C.POINTER(T.PoaParams), C.POINTER(T.CnsOut)]
        L.hx_poa_supports.argtypes = [C.c_void_p, C.POINTER(T.CoordsOut), C.POINTER(T.PoaParams), C.POINTER(T.CnsOut)]
        L.hx_poa_sequences.argtypes = [C.c_void_p, C.c_uint32, T.u64p, T.u64p, C.c_char_p, C.POINTER(T.PoaParams), C.POINTER(T.CnsOut)]
        L.hx_free_chain.argtypes = [C.c_void_p, C.POINTER(T.ChainOut)]
        L.hx_free_edges.argtypes = [C.c_void_p, C.POINTER(T.EdgesOut)]
        L.hx_free_coords.argtypes = [C.c_void_p, C.POINTER(T.CoordsOut)]
        L.hx_free_cns.argtypes = [C.c_void_p, C.POINTER(T.CnsOut)]
        L.hx_edge_emit.argtypes = [C.c_void_p, C.POINTER(T.Params), C.POINTER(C.c_uint64)]
        L.hx_edge_records_bytes.restype = C.c_uint32
        L.hx_edge_records_export.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.hx_edge_records_import.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(T.EdgesOut)]
        L.hx_timing_reset.argtypes = [C.c_void_p]
        L.hx_timing_get.argtypes = [C.c_void_p, C.POINTER(C.c_double * 4), C.POINTER(C.c_uint64 * 4)]
        L.hx_set_poa_block.argtypes = [C.c_void_p, C.c_int]
        L.hx_set_poa_traceback.argtypes = [C.c_void_p, C.c_int]
        L.hx_poa_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 6), C.POINTER(C.c_uint64 * 6)]
        L.hx_poa_phase_cycles.restype = C.c_uint32
        L.hx_backend_fill.argtypes = [C.c_void_p, C.POINTER(T.Backend)]
        L.hx_poa_workspace_bytes.argtypes = [C.c_void_p]
        L.hx_poa_workspace_bytes.restype = C.c_uint64
        # multi-GPU inside one process (one thread per rank): hx_group_*
        L.hx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.hx_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]
        L.hx_option_names.restype = C.c_char_p
        L.hx_poa_memory_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.hx_poa_release_workspace.argtypes = [C.c_void_p]
        L.hx_poa_reserve.argtypes = [C.c_void_p, C.c_uint64]
        L.hx_poa_host_times.argtypes = [C.c_void_p, C.POINTER(C.c_double * 8)]
        L.hx_poa_host_times.restype = None
        L.hx_poa_arena_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.hx_poa_arena_stats.restype = None
        L.hx_poa_prune_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 4)]
        L.hx_group_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_char_p, C.POINTER(C.c_void_p)]
        L.hx_group_set_timeout.argtypes = [C.c_void_p, C.c_double]
        L.hx_group_set_timeout.restype = None
        L.hx_group_inject_fault.argtypes = [C.c_void_p, C.c_int]
        L.hx_group_inject_fault.restype = None
        L.hx_group_destroy.argtypes = [C.c_void_p]
        L.hx_group_size.argtypes = [C.c_void_p]
        L.hx_group_ctx.argtypes = [C.c_void_p, C.c_int]
        L.hx_group_ctx.restype = C.c_void_p
        L.hx_group_transport.argtypes = [C.c_void_p]
        L.hx_group_transport.restype = C.c_char_p
        L.hx_edge_merge.argtypes = [C.c_void_p, C.c_int, C.POINTER(T.Params), C.POINTER(T.EdgesOut)]
        L.hx_group_backend_fill.argtypes = [C.c_void_p, C.c_int, C.POINTER(T.Backend)]
        L.hx_group_exchange_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.hx_group_rccl_ranks.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _lib = L
    return _lib


def option_names():
    return lib().hx_option_names().decode().split(",")


def env_options(environ=None):
    """the HX_* variables of the environment that name library options ({option: text}): what an APPLICATION hands to hx_set_option when it
    creates a context - the library itself reads no environment (include/haslr_hip.h)"""
    environ = os.environ if environ is None else environ
    names = set(option_names()) | {"prof1", "prof2", "prof3"}
    return {k[3:].lower(): v for k, v in environ.items() if k.startswith("HX_") and k[3:].lower() in names}


class HipContext:
    """One GPU: resident inputs + the four hot-path operators."""

    def __init__(self, device=0, stream=None, options=None, use_env=True):
        L = lib()
        h = C.c_void_p()
        if L.hx_ctx_create(device, stream, C.byref(h)) != 0:
            raise HipError(L.hx_last_error().decode())
        self._h = h
        self._ds = None
        self.table = T.Backend()
        L.hx_backend_fill(self._h, C.byref(self.table))
        try:
            if use_env:   # (this Python process is the application: its HX_* variables are the context's options, copied ONCE; a malformed value is
                          # reported and ignored - explicit set_option calls stay strict)
                for k, v in env_options().items():
                    try:
                        self.set_option(k, v)
                    except HipError as e:
                        import warnings
                        warnings.warn(f"HX_{k.upper()}={v!r} ignored: {e}")
            if options:
                self.set_options(**options)
        except Exception:
            self.close()
            raise

    def set_option(self, name, value):
        """tuning / test switch of this context (include/haslr_hip.h: hx_set_option); value None = back to the default"""
        self._chk(lib().hx_set_option(self._h, str(name).encode(), None if value is None else str(value).encode()))

    def set_options(self, **kv):
        for k, v in kv.items():
            self.set_option(k, v)

    def get_option(self, name):
        v = C.c_double()
        self._chk(lib().hx_get_option(self._h, str(name).encode(), C.byref(v)))
        return v.value

    def options(self, **kv):
        """context manager: the options hold inside the block and return to what they were afterwards"""
        ctx = self

        class _Scope:
            def __enter__(self_):
                self_.old = {k: ctx.get_option(k) for k in kv}
                ctx.set_options(**kv)
                return ctx

            def __exit__(self_, *exc):
                for k, v in self_.old.items():
                    ctx.set_option(k, int(v) if float(v).is_integer() else v)
                return False
        return _Scope()

    def poa_memory_stats(self):
        a, b, w = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().hx_poa_memory_stats(self._h, C.byref(a), C.byref(b), C.byref(w))
        return {"free_at_first_call": a.value, "budget": b.value, "last_call_workspace": w.value}

    def poa_release_workspace(self):
        self._chk(lib().hx_poa_release_workspace(self._h))

    def poa_reserve(self, nbytes):
        """the consensus workspace's arena, ahead of the first call (include/haslr_hip.h: hx_poa_reserve)"""
        self._chk(lib().hx_poa_reserve(self._h, int(nbytes)))

    def poa_host_times(self):
        o = (C.c_double * 8)()
        lib().hx_poa_host_times(self._h, C.byref(o))
        return {"plan_ms": o[0], "workspace_ms": o[1], "enqueue_ms": o[2], "device_wait_ms": o[3], "collect_ms": o[4], "finish_ms": o[5], "total_ms": o[7]}

    def poa_arena_stats(self):
        cap, n, ms = C.c_uint64(), C.c_uint64(), C.c_double()
        lib().hx_poa_arena_stats(self._h, C.byref(cap), C.byref(n), C.byref(ms))
        return {"bytes": cap.value, "allocations": n.value, "alloc_ms": ms.value}

    def poa_prune_stats(self):
        o = (C.c_uint64 * 4)()
        lib().hx_poa_prune_stats(self._h, C.byref(o))
        return {"wave_rows": o[0], "wave_rows_skipped": o[1], "attempts_repeated": o[2], "alignments_with_threshold": o[3]}

    def _chk(self, rc):
        if rc != 0:
            raise HipError(lib().hx_last_error().decode())

    def upload(self, dataset):
        self._ds = dataset
        self._chk(lib().hx_upload(self._h, C.byref(dataset.contigs), C.byref(dataset.reads), C.byref(dataset.hits), dataset.read_hit_off))
        lib().hx_set_prefiltered(self._h, int(getattr(dataset, "used_longread_index", False)))   # records of an index.longread are taken as they are

    def set_read_shard(self, b, e):
        self._chk(lib().hx_set_read_shard(self._h, b, e))

    def set_poa_block(self, threads):
        lib().hx_set_poa_block(self._h, threads)

    def set_poa_traceback(self, use_direction_bytes):
        lib().hx_set_poa_traceback(self._h, int(use_direction_bytes))

    def backend(self):
        return self.table

    # ---- direct operator calls (tests); results are returned as dicts of numpy arrays
    def chain_reads(self, params):
        o = T.ChainOut()
        self._chk(lib().hx_chain_reads(self._h, C.byref(params), C.byref(o)))
        d = T.chain_to_dict(o)
        lib().hx_free_chain(self._h, C.byref(o))
        return d

    def edge_support(self, params, sides=True):
        o = T.EdgesOut()
        self._chk(lib().hx_edge_support(self._h, C.byref(params), C.byref(o)))
        d = T.edges_to_dict(o, sides)
        lib().hx_free_edges(self._h, C.byref(o))
        return d

    def edge_emit(self, params):
        n = C.c_uint64()
        self._chk(lib().hx_edge_emit(self._h, C.byref(params), C.byref(n)))
        return n.value

    def edge_records_export(self, dst_ptr, capacity):
        self._chk(lib().hx_edge_records_export(self._h, dst_ptr, capacity))

    def edge_records_import(self, src_ptr, n, sides=True):
        o = T.EdgesOut()
        self._chk(lib().hx_edge_records_import(self._h, src_ptr, n, C.byref(o)))
        d = T.edges_to_dict(o, sides)
        lib().hx_free_edges(self._h, C.byref(o))
        return d

    def edge_coords(self, sel):
        import numpy as np
        sel = np.ascontiguousarray(sel, dtype=np.uint32)
        o = T.CoordsOut()
        self._chk(lib().hx_edge_coords(self._h, len(sel), sel.ctypes.data_as(T.u32p), C.byref(o)))
        d = T.coords_to_dict(o)
        lib().hx_free_coords(self._h, C.byref(o))
        return d

    def poa_batch(self, match=5, mismatch=-4, gap=-8):
        o = T.CnsOut()
        pp = T.PoaParams(match, mismatch, gap)
        self._chk(lib().hx_poa_batch(self._h, C.byref(pp), C.byref(o)))
        r = T.cns_to_list(o), {"dp_cells": o.dp_cells, "seq_bases": o.seq_bases, "n_aligned": o.n_aligned}
        lib().hx_free_cns(self._h, C.byref(o))
        return r

    def poa_supports(self, supports, match=5, mismatch=-4, gap=-8):
        """consensus of caller-given edges: supports = list (one per edge) of (read id, strand, spos, epos) tuples into the resident reads"""
        import numpy as np
        off = np.zeros(len(supports) + 1, dtype=np.uint64)
        flat = [t for e in supports for t in e]
        for i, e in enumerate(supports):
            off[i + 1] = off[i] + len(e)
        lr = np.array([r | (s << 31) for r, s, _, _ in flat] or [0], dtype=np.uint32)
        sp = np.array([t[2] & 0xffffffff for t in flat] or [0], dtype=np.uint32)
        ep = np.array([t[3] & 0xffffffff for t in flat] or [0], dtype=np.uint32)
        sup = T.CoordsOut(len(supports), None, None, off.ctypes.data_as(T.u64p), lr.ctypes.data_as(T.u32p), sp.ctypes.data_as(T.u32p), ep.ctypes.data_as(T.u32p))
        o, pp = T.CnsOut(), T.PoaParams(match, mismatch, gap)
        self._chk(lib().hx_poa_supports(self._h, C.byref(sup), C.byref(pp), C.byref(o)))
        r = T.cns_to_list(o)
        lib().hx_free_cns(self._h, C.byref(o))
        return r

    def poa_sequences(self, sets, match=5, mismatch=-4, gap=-8):
        """consensus of every set of plain ACGT strings (aligned in the given order); nothing has to be resident"""
        import numpy as np
        set_off = np.zeros(len(sets) + 1, dtype=np.uint64)
        seqs = [q for st in sets for q in st]
        for i, st in enumerate(sets):
            set_off[i + 1] = set_off[i] + len(st)
        seq_off = np.zeros(len(seqs) + 1, dtype=np.uint64)
        for i, q in enumerate(seqs):
            seq_off[i + 1] = seq_off[i] + len(q)
        o, pp = T.CnsOut(), T.PoaParams(match, mismatch, gap)
        self._chk(lib().hx_poa_sequences(self._h, len(sets), set_off.ctypes.data_as(T.u64p), seq_off.ctypes.data_as(T.u64p), "".join(seqs).encode(), C.byref(pp), C.byref(o)))
        r = T.cns_to_list(o)
        lib().hx_free_cns(self._h, C.byref(o))
        return r

    def poa_phase_cycles(self):
        a, b = (C.c_uint64 * 6)(), (C.c_uint64 * 6)()
        n = lib().hx_poa_phase_cycles(self._h, C.byref(a), C.byref(b))
        names = ("decode", "dp", "traceback", "graph_update", "toposort", "csr")
        return {"edges": n, "sum": dict(zip(names, a)), "slowest_edge": dict(zip(names, b))}

    def poa_workspace_bytes(self):
        return int(lib().hx_poa_workspace_bytes(self._h))

    def timing_reset(self):
        lib().hx_timing_reset(self._h)

    def timing(self):
        ms, n = (C.c_double * 4)(), (C.c_uint64 * 4)()
        lib().hx_timing_get(self._h, C.byref(ms), C.byref(n))
        names = ("chain", "edges", "coords", "poa")
        return {k: {"ms": ms[i], "launches": n[i]} for i, k in enumerate(names)}

    def close(self):
        if self._h:
            lib().hx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def records_bytes():
    return lib().hx_edge_records_bytes()
