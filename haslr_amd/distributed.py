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


def allgather_records(local: torch.Tensor, n_local: int, rec_bytes: int, group=None):
    """All-gather of variable-length packed record buffers (uint8 tensors of n_local*rec_bytes bytes).
    Returns (merged tensor in rank order, total record count). One data collective (+ a count exchange):
    every rank contributes its buffer padded to the largest, the padding is cut out afterwards."""
    world = dist.get_world_size(group)
    dev = local.device
    mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, mine, group=group)
    counts_h = [int(c.item()) for c in counts]
    cap = max(max(counts_h), 1) * rec_bytes
    padded = torch.zeros(cap, dtype=torch.uint8, device=dev)
    padded[: n_local * rec_bytes] = local[: n_local * rec_bytes]
    gathered = torch.empty(world * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    if sum(counts_h) == 0:
        return torch.zeros(0, dtype=torch.uint8, device=dev), 0
    if all(c == counts_h[0] for c in counts_h):
        return gathered, int(sum(counts_h))                       # no padding anywhere: the gather buffer is the result
    merged = torch.cat([gathered[r * cap: r * cap + c * rec_bytes] for r, c in enumerate(counts_h)])
    return merged.contiguous(), int(sum(counts_h))


def allgather_bytes(blob: bytes, device, group=None) -> bytes:
    """Every rank's byte string, concatenated in rank order."""
    local = torch.frombuffer(bytearray(blob) if blob else bytearray(1), dtype=torch.uint8).to(device)
    merged, total = allgather_records(local, len(blob), 1, group)
    return merged.cpu().numpy().tobytes()[:total]


def agree(ok: bool, device, group=None, what="stage"):
    """Every rank arrives with its own verdict, every rank leaves with the worst one: a rank that failed never leaves the others waiting in
    the next collective (one 1-element all-reduce). Raises on every rank when any rank failed."""
    t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    if int(t.item()):
        raise RuntimeError(f"{what} failed on {'this rank' if not ok else 'another rank'} (all ranks stop)")


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

    def _edge_support(self, _ctx, _prm, out):
        try:
            n, local = 0, None
            try:
                n = self.records.emit()
                local = self.records.export(n)
            except Exception as e:  # noqa: BLE001
                self.error = e
            agree(local is not None, getattr(self.records, "comm_device", torch.device("cpu")), self.group, "edge-record emission")   # before the all-gather: all or nobody
            t0 = time.perf_counter()
            merged, total = allgather_records(local, n, self.records.rec_bytes, self.group)
            if merged.is_cuda:
                torch.cuda.synchronize()
            self.exchange_ms = (time.perf_counter() - t0) * 1e3       # counts exchange + the padded all-gather of the records (+ the cut of the padding)
            self.exchange_bytes = total * self.records.rec_bytes
            return self.records.import_(merged, total, out)
        except Exception as e:  # noqa: BLE001 - must not propagate through the C callback
            self.error = e
            print(f"[ERROR] sharded edge_support: {e}", flush=True)
            return -1


def gather_results(run, device, group=None):
    """All ranks exchange the coordinates / supports / consensus of their share of the edges; afterwards every
    rank's run holds all of them and can stitch. Returns the number of bytes gathered."""
    blob, err = None, None
    try:
        blob = run.results_export()
    except Exception as e:  # noqa: BLE001
        err = e
    agree(blob is not None, device, group, f"results export ({err})" if err else "results export")
    merged = allgather_bytes(blob, device, group)
    run.results_import(merged)
    if run.results_missing:
        raise RuntimeError(f"{run.results_missing} edges are without results after the gather")
    return len(merged)


def sharded_stages(run, device, group=None):
    """chain -> graph (the record all-gather happens inside, behind its own agreement) -> coords -> consensus -> gathered results, with the ranks
    agreeing on success after every stage (no rank is left waiting in a later collective). Returns the bytes of results gathered."""
    def stage(fn, name):
        err = None
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            err = e
        try:
            agree(err is None, device, group, name)
        except RuntimeError as a:
            raise (err or a)
    stage(run.chain, "chain stage")
    # the record all-gather inside fails on every rank together (agreement in the backend's edge_support); what follows it on a rank - the
    # import, the graph build, rank 0's GFA / stat / log files - can still fail alone, so the stage as a whole is agreed on as well
    stage(run.graph, "graph stage")
    stage(lambda: (run.coords(), run.consensus()), "coordinate / consensus stage")
    return gather_results(run, device, group)


def run_sharded(ds, params, backend: ShardedBackend, lr_begin, rank, world, device, group=None, out_dir=None, assemble=True):
    """One pass of the stage on this rank: chain (own reads) -> merged graph -> coordinates + consensus (own edges)
    -> gathered results -> assembly. Rank 0 passes out_dir to get the reference's output files."""
    from . import host
    run = host.Run(ds, params, backend.table, out_dir)
    run.set_edge_shard(rank, world)
    run.set_read_shard(lr_begin)
    sharded_stages(run, device, group)
    if out_dir is not None or world > 1:
        text = allgather_bytes(run.compact_text(), device, group)   # compact_uniq.txt lists every read: rank order = read order
        if out_dir is not None:
            with open(f"{out_dir}/compact_uniq.txt", "wb") as f:
                f.write(text)
    if assemble:
        run.assemble()
    return run
