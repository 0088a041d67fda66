"""Multi-GPU orchestration: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The path shards twice with ONE exchange of inputs in between:
  phase 1  long reads are split into contiguous id ranges balanced on raw-hit count; every rank filters /
           sorts / trims / chains its reads and emits their edge-support records
  exchange one all-gather of the packed records (ranks own ascending read ranges, so concatenation in rank
           order followed by the stable key sort reproduces the reference's per-edge support order)
  phase 2  every rank sorts + segments the full multiset and cleans the (small) graph redundantly, then
           computes coordinates and POA consensus for its share of the surviving edges (dealt by estimated
           DP cost, haslr_host.h hxh_run_set_edge_shard)
  results  one all-gather of the per-edge results (coordinates, supports, consensus strings): every rank
           can then stitch the assembly (asm_get_assembly needs all of them, Assemble.cpp:1045-1077)
Raw inputs (CIGAR ops, packed reads) are replicated on every GPU, so records only carry indices.

Collectives of one pass (round 4: four, two of them on the data path - what the in-binary group of `haslr_assemble --gpus N` does with one
count exchange through the process's memory and one ncclAllGather each way): before each of the two all-gathers ONE small all-gather carries
every rank's byte count AND its verdict on everything it did since the previous exchange (`exchange_counts`), so a rank that failed takes every
rank out together and nobody is left waiting in a collective - without separate agreement rounds.

`ShardedBackend` is the compute-backend table of a rank; `run_sharded` drives one whole pass. Both take the
record source as an object with emit() / export(buffer) / import_(buffer, n, out), so that the CPU tests can
drive exactly this code over gloo with the test oracle in place of the HIP context.
"""
import ctypes as C
import time

import numpy as np
import torch
import torch.distributed as dist

from . import ctypes_defs as T


def shard_bounds(read_hit_off, n_reads, world):
    """Contiguous read-id ranges with ~equal numbers of raw PAF records. Returns world+1 boundaries."""
    rho = np.ctypeslib.as_array(read_hit_off, shape=(n_reads + 1,))
    total = int(rho[-1])
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(rho, total * r // world, side="left")))
    bounds.append(n_reads)
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def exchange_counts(values, ok, device, group=None, what="stage"):
    """ONE small all-gather: every rank's verdict (`ok`) on what it did since the last exchange, and its counts for the data collective that
    follows. Every rank leaves with all ranks' counts - or, when any rank arrived with a failure, with the same RuntimeError (a rank that
    failed never leaves the others waiting in the next collective). One host synchronisation."""
    world = dist.get_world_size(group)
    mine = torch.tensor([0 if ok else 1] + [int(v) for v in values], dtype=torch.int64, device=device)
    out = torch.empty(world * mine.numel(), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine, group=group)
    rows = out.cpu().view(world, -1).tolist()
    bad = [r for r, row in enumerate(rows) if row[0]]
    if bad:
        raise RuntimeError(f"{what} failed on rank{'s' if len(bad) > 1 else ''} {', '.join(map(str, bad))} ({'another rank' if ok else 'this rank among them'}): all ranks stop")
    return [row[1:] for row in rows]


def allgather_padded(local: torch.Tensor, counts_bytes, group=None):
    """All-gather of byte buffers whose lengths every rank already knows (`counts_bytes`, from exchange_counts): every rank contributes its
    buffer padded to the largest, the padding is cut out afterwards (the gather buffer is the result when there is none). ONE collective."""
    world = dist.get_world_size(group)
    dev = local.device
    n_local = counts_bytes[dist.get_rank(group)]
    cap = max(max(counts_bytes), 1)
    if local.numel() == cap:
        padded = local
    else:
        padded = torch.zeros(cap, dtype=torch.uint8, device=dev)
        padded[:n_local] = local[:n_local]
    gathered = torch.empty(world * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    if sum(counts_bytes) == 0:
        return torch.zeros(0, dtype=torch.uint8, device=dev)
    if all(c == cap for c in counts_bytes):
        return gathered
    return torch.cat([gathered[r * cap: r * cap + c] for r, c in enumerate(counts_bytes)]).contiguous()


def allgather_records(local: torch.Tensor, n_local: int, rec_bytes: int, group=None):
    """All-gather of variable-length packed record buffers (uint8 tensors of n_local*rec_bytes bytes).
    Returns (merged tensor in rank order, total record count): the count exchange, then one data collective."""
    counts = [c[0] for c in exchange_counts([n_local], True, local.device, group, "record exchange")]
    merged = allgather_padded(local, [c * rec_bytes for c in counts], group)
    return merged, int(sum(counts))


def allgather_bytes(blob: bytes, device, group=None) -> bytes:
    """Every rank's byte string, concatenated in rank order."""
    local = torch.frombuffer(bytearray(blob) if blob else bytearray(1), dtype=torch.uint8).to(device)
    merged, total = allgather_records(local, len(blob), 1, group)
    return merged.cpu().numpy().tobytes()[:total]


class HipRecords:
    """Edge-support records of a rank's read shard on its HIP context (include/haslr_hip.h: hx_edge_emit / _export / _import)."""

    def __init__(self, ctx, params, comm_device=None):
        """comm_device: where the collective runs (the GPU for RCCL; torch.device("cpu") stages through host memory, for gloo)"""
        from . import hip
        self.ctx, self.params = ctx, params
        self.rec_bytes = hip.records_bytes()
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.comm_device = comm_device or self.device

    def emit(self):
        return self.ctx.edge_emit(self.params)

    def export(self, n):
        local = torch.empty(max(n, 1) * self.rec_bytes, dtype=torch.uint8, device=self.device)
        self.ctx.edge_records_export(C.c_void_p(local.data_ptr()), n)
        return local.to(self.comm_device)

    def import_(self, merged, total, out):
        from . import hip
        merged = merged.to(self.device)
        torch.cuda.synchronize()
        return hip.lib().hx_edge_records_import(self.ctx._h, C.c_void_p(merged.data_ptr()), total, out)


class ShardedBackend:
    """Backend table for the host pipeline in a multi-GPU run: chain/coords/POA are the rank's own operators
    (restricted to its read shard / its share of the edges); edge_support = emit + all-gather + import."""

    def __init__(self, table, records, group=None):
        self.records, self.group = records, group
        self.table = T.Backend()
        C.memmove(C.byref(self.table), C.byref(table), C.sizeof(T.Backend))
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(T.Params), C.POINTER(T.EdgesOut))
        self._cb = proto(self._edge_support)
        self.table.edge_support = C.cast(self._cb, C.c_void_p).value
        self.exchange_bytes = 0
        self.exchange_ms = 0.0
        self.error = None
        self.pending_error = None
        self.stopped = None      # the RuntimeError of a count round that stopped all ranks
        self.exchanged = False   # this pass's record exchange (its count round) has been entered: sharded_stages resets it per pass

    def _edge_support(self, _ctx, _prm, out):
        try:
            self.exchanged = True
            n, local, err = 0, None, self.pending_error
            if err is None:
                try:
                    n = self.records.emit()
                    local = self.records.export(n)
                except Exception as e:  # noqa: BLE001
                    err = self.error = e
            dev = getattr(self.records, "comm_device", torch.device("cpu"))
            t0 = time.perf_counter()
            # the counts and the verdicts on chain + emission travel together: all ranks go into the all-gather, or none
            try:
                counts = [c[0] for c in exchange_counts([n], err is None, dev, self.group, "chain stage / edge-record emission")]
            except RuntimeError as stop:
                self.stopped = stop
                raise
            merged = allgather_padded(local, [c * self.records.rec_bytes for c in counts], self.group)
            total = int(sum(counts))
            if merged.is_cuda:
                torch.cuda.synchronize()
            self.exchange_ms = (time.perf_counter() - t0) * 1e3       # counts exchange + the padded all-gather of the records (+ the cut of the padding)
            self.exchange_bytes = total * self.records.rec_bytes
            return self.records.import_(merged, total, out)
        except Exception as e:  # noqa: BLE001 - must not propagate through the C callback
            self.error = self.error or e
            print(f"[ERROR] sharded edge_support: {e}", flush=True)
            return -1

    def fail_exchange(self, err):
        """A rank whose chain stage failed still takes part in the record exchange's count round - with its verdict - so that every rank stops there."""
        self.pending_error = err
        try:
            self._edge_support(None, None, None)
        finally:
            self.pending_error = None


def gather_results(run, device, group=None, err=None, text=None):
    """All ranks exchange the coordinates / supports / consensus of their share of the edges (and, `text` given, their part of
    compact_uniq.txt in the same buffer); afterwards every rank's run holds all of them and can stitch. `err`: what went wrong on this rank
    since the record exchange (it then arrives with that verdict and every rank raises). Returns (bytes of results gathered, merged text)."""
    blob = None
    if err is None:
        try:
            blob = run.results_export()
        except Exception as e:  # noqa: BLE001
            err = e
    tbytes = text if (text is not None and err is None) else b""
    try:
        counts = exchange_counts([len(blob or b""), len(tbytes)], err is None, device, group, "graph / coordinate / consensus stage")
    except RuntimeError as a:
        raise (err or a)
    mine = (blob or b"") + tbytes
    local = torch.frombuffer(bytearray(mine) if mine else bytearray(1), dtype=torch.uint8).to(device)
    merged = allgather_padded(local, [c[0] + c[1] for c in counts], group).cpu().numpy().tobytes()
    res, txt, o = [], [], 0
    for nres, ntxt in counts:
        res.append(merged[o: o + nres]); txt.append(merged[o + nres: o + nres + ntxt]); o += nres + ntxt
    res = b"".join(res)
    run.results_import(res)
    if run.results_missing:
        raise RuntimeError(f"{run.results_missing} edges are without results after the gather")
    return len(res), b"".join(txt)


def sharded_stages(run, device, group=None, backend=None, with_text=False):
    """chain -> graph (the record exchange happens inside) -> coords -> consensus -> gathered results: two exchanges, each a count round that
    also carries the ranks' verdicts on everything since the previous one, and one all-gather. Returns the bytes of results gathered
    (with_text: and the merged compact_uniq text). `backend` (the rank's ShardedBackend) is required: a rank that fails before the record
    exchange must still enter its count round - with its verdict - or the other ranks wait in the collective."""
    if backend is None:
        raise TypeError("sharded_stages needs the rank's ShardedBackend (failure agreement of the record exchange goes through it)")
    backend.exchanged = False
    backend.stopped = None
    backend.error = None                   # (the first error of THIS pass is the one a count round reports, not one left over from an earlier pass)
    if hasattr(backend, "pending_error"):
        backend.pending_error = None
    err = None
    try:
        run.chain()
    except Exception as e:  # noqa: BLE001
        err = e
    if err is not None:
        backend.fail_exchange(err)        # this rank's verdict reaches the count round of the record exchange: every rank stops there
        raise err
    # the record exchange inside fails on every rank together; what follows it on a rank - the import, the graph build, rank 0's GFA / stat /
    # log files - can still fail alone: that verdict travels with the count round of the results exchange
    try:
        run.graph()
    except Exception as e:  # noqa: BLE001
        if backend.stopped is not None:
            raise (backend.error if backend.error is not None and backend.error is not backend.stopped else backend.stopped)   # every rank is raising here
        if not backend.exchanged:
            # the graph stage failed BEFORE it reached the record exchange: the other ranks are in (or on their way to) that count round, not the
            # results one - this rank joins it with its verdict, and every rank stops there
            backend.fail_exchange(e)
            raise
        err = e
    text = None
    if err is None:
        try:
            run.coords()
            run.consensus()
            if with_text:
                text = run.compact_text()      # (inside the try: a failure here travels as this rank's verdict like any other)
        except Exception as e:  # noqa: BLE001
            err = e
    n, text = gather_results(run, device, group, err, text if (with_text and err is None) else (b"" if with_text else None))
    return (n, text) if with_text else n


def run_sharded(ds, params, backend: ShardedBackend, lr_begin, rank, world, device, group=None, out_dir=None, assemble=True):
    """One pass of the stage on this rank: chain (own reads) -> merged graph -> coordinates + consensus (own edges)
    -> gathered results -> assembly. Rank 0 passes out_dir to get the reference's output files."""
    from . import host
    run = host.Run(ds, params, backend.table, out_dir)
    run.set_edge_shard(rank, world)
    run.set_read_shard(lr_begin)
    _, text = sharded_stages(run, device, group, backend, with_text=True)   # compact_uniq.txt lists every read: rank order = read order
    if out_dir is not None:
        with open(f"{out_dir}/compact_uniq.txt", "wb") as f:
            f.write(text)
    if assemble:
        run.assemble()
    return run
