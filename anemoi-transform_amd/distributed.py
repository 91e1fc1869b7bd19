"""Multi-GPU regrid: target points sharded over ranks, source stack sent once.

One process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on
MI355X, ``gloo`` in the CPU tests).  The reference has no counterpart: it is a
single-process loop.  The partitioning follows SURVEY.md §8e:

* row ``t`` of the interpolation operator reads only source points and writes only
  target ``t``, so ``[0, n_tgt)`` is cut into ``world`` contiguous slices
  (``GatherPlan.shard``; boundaries balanced by HBM traffic, not by count) and the slices never talk
  to each other;
* every rank needs the source stack: it is broadcast ONCE (``broadcast_stack``,
  one RCCL broadcast of the contiguous stack tensor) — or, better, each rank only
  receives the band of source columns its slice references (``source_band``,
  ``exchange_source_bands``): in the column layout that band is one contiguous slab;
* outputs stay sharded (``FieldList`` per rank); ``gather_target_shards``
  assembles the full field on every rank for callers that need it.

Two transports, same functions: ``torch.distributed`` (default; ``comm=None``), or the library's own C-ABI
communicator (``native.Comm`` = ``atx_comm_*`` of include/atx.h, RCCL bound directly) passed as ``comm=`` —
the route a ctypes-only binder of the reference would take.  ``atx_comm_from_torch`` creates
one inside a torch.distributed job (the unique id travels through the job's store).
"""

from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from .gather import GatherPlan
from .stack import COLUMNS, Stack


def init_process_group(backend: str | None = None) -> tuple[int, int]:
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; returns (rank, world)."""
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this platform
        kwargs = {}
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            kwargs["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kwargs)
    return dist.get_rank(), dist.get_world_size()


_data_group = None  # process group that carries stacks (None = the default group)


def set_data_group(group) -> None:
    """Carry the stack traffic of this module on ``group`` (e.g. an ``nccl`` group = RCCL over xGMI) while the job's default
    group stays a host-side one (``gloo``) for barriers and small reductions — how bench.py runs, so that its timing
    skeleton does not depend on the collective library it measures."""
    global _data_group
    _data_group = group


def warm_up_transport() -> None:
    """One-element messages over every route the stack traffic will take on the data group — a broadcast from every rank and
    a send/recv between every pair — so that the collective library's set-up (RCCL: communicator, rings, point-to-point
    channels) is not billed to the first real exchange."""
    rank, world = dist.get_rank(), dist.get_world_size()
    device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    token = torch.zeros(1, dtype=torch.float32, device=device)
    for r in range(world):
        dist.broadcast(token, src=r, group=_data_group)
    inbox = [torch.zeros(1, dtype=torch.float32, device=device) for _ in range(world)]
    ops = []
    for r in range(world):
        if r != rank:
            ops.append(dist.P2POp(dist.isend, token, r, group=_data_group))
            ops.append(dist.P2POp(dist.irecv, inbox[r], r, group=_data_group))
    if ops:
        for work in dist.batch_isend_irecv(ops):
            work.wait()


def atx_comm_from_torch():
    """A ``native.Comm`` (RCCL through the C ABI) spanning the ranks of the initialised torch.distributed job: rank 0
    draws the unique id, the job's own transport (any backend) hands it round."""
    from . import native

    rank, world = dist.get_rank(), dist.get_world_size()
    box: list = [None]
    if rank == 0:
        try:
            box[0] = native.Comm.unique_id()
        except Exception as e:  # every rank must learn of it, or the others would wait for the id forever
            box[0] = f"{type(e).__name__}: {e}"
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    if not isinstance(box[0], bytes):
        raise native.AtxError(f"rank 0 could not create an RCCL unique id: {box[0]}")
    return native.Comm(world, rank, box[0])


def _rank_world(comm) -> tuple[int, int]:
    if comm is not None:
        return comm.rank, comm.world
    return dist.get_rank(), dist.get_world_size()


def broadcast_stack(stack: Stack | None, src: int, *, n_pts: int, n_lev: int, dtype: torch.dtype, device: torch.device,
                    layout: int = COLUMNS, comm=None) -> Stack:
    """The source stack of rank ``src`` on every rank (one broadcast of the whole pitched tensor)."""
    rank, _ = _rank_world(comm)
    if rank == src:
        assert stack is not None and (stack.n_pts, stack.n_lev, stack.layout) == (n_pts, n_lev, layout)
        buf = stack
    else:
        buf = Stack.empty(n_pts, n_lev, dtype, device, layout)
    if comm is not None:
        comm.bcast(buf.data, src)
    else:
        dist.broadcast(buf.data, src=src, group=_data_group)
    return buf


def exchange_stacks(mine: Stack, comm=None, collective: str = "broadcast") -> list[Stack]:
    """Every rank contributes one stack of identical shape; every rank ends up with all of them
    (the "source broadcast once" step of a target-sharded job).

    ``collective="broadcast"`` (default): ``world`` broadcasts, one per contributing rank, each usable as soon as it has landed (what the
    double-buffered step pipelines).  ``collective="all_gather"``: ONE all-gather (``all_gather_into_tensor`` / ``atx_all_gather``) into a single buffer of
    ``world`` stacks — on a fully connected xGMI node every link carries its share at once, where a chain of broadcasts is paced by the
    root's route each time; the stacks returned are views of that buffer."""
    rank, world = _rank_world(comm)
    if collective == "all_gather":
        rows = mine.data.shape[0]  # (concatenated along the rows: the output form every backend accepts)
        buf = torch.empty((world * rows, mine.data.shape[1]), dtype=mine.data.dtype, device=mine.data.device)
        if comm is not None:
            comm.all_gather(mine.data.contiguous(), buf)  # atx_all_gather: ncclAllGather through the C ABI
        else:
            dist.all_gather_into_tensor(buf, mine.data.contiguous(), group=_data_group)
        return [Stack(buf[r * rows:(r + 1) * rows], mine.n_pts, mine.n_lev, mine.layout) for r in range(world)]
    if collective != "broadcast":
        raise ValueError(f"collective must be 'broadcast' or 'all_gather', got {collective!r}")
    out = []
    for r in range(world):
        out.append(broadcast_stack(mine if r == rank else None, r, n_pts=mine.n_pts, n_lev=mine.n_lev, dtype=mine.dtype,
                                   device=mine.device, layout=mine.layout, comm=comm))
    return out


def source_band(plan: GatherPlan) -> tuple[int, int]:
    """``[lo, hi)`` range of source points a (sharded) plan references.  For a latitude-ordered
    target slice this is a narrow band of the source grid — with column stacks one contiguous
    slab of HBM, so a rank can be fed its band instead of the whole stack."""
    idx = plan.index if plan.kind == "ell" else plan.indices
    idx = idx[idx >= 0]  # padded rows mark absent entries with -1
    if idx.size == 0:
        return 0, 0
    return int(idx.min()), int(idx.max()) + 1


def rebase_plan(plan: GatherPlan, lo: int, hi: int) -> GatherPlan:
    """The same plan expressed against the source slab ``[lo, hi)``."""
    if plan.kind == "ell":
        index = plan.index.astype(np.int64)
        based = GatherPlan(hi - lo, plan.n_tgt, index=np.where(index >= 0, index - lo, -1), weights=plan.weights, padded=plan.padded)
    else:
        based = GatherPlan(hi - lo, plan.n_tgt, csr=(plan.data, plan.indices.astype(np.int64) - lo, plan.indptr))
    based.order = plan.order  # the same targets, visited in the same order
    return based


def _band_layout(plan: GatherPlan, rank: int, world: int) -> tuple[list[tuple[int, int]], GatherPlan]:
    """``(source band of every rank's target slice, this rank's slice rebased onto its own band)`` — host work over the whole index
    table (17 ms for O1280 -> 0.25 degree), remembered on the plan: a job exchanges bands at every step with the same plan."""
    cached = plan.__dict__.setdefault("_bands", {})
    key = (int(rank), int(world))
    if key not in cached:
        ranges = [source_band(plan.shard(r, world)) for r in range(world)]  # same on every rank: the plan is replicated
        cached[key] = (ranges, rebase_plan(plan.shard(rank, world), *ranges[rank]))
    return cached[key]


def exchange_source_bands(mine: Stack, plan: GatherPlan, comm=None) -> tuple[list[Stack], GatherPlan]:
    """Band-limited source exchange: every rank contributes one source stack and receives, from every
    rank, only the slab of source columns its own target slice references.

    Returns ``(bands, local_plan)``: ``bands[r]`` is rank ``r``'s stack restricted to this rank's band and
    ``local_plan`` is this rank's shard of ``plan`` rebased onto that band, so
    ``local_plan.apply(bands[r])`` equals ``plan.shard(rank, world).apply(<rank r's full stack>)`` bit for bit.
    Point-to-point (RCCL send/recv over xGMI): ``world - 1`` slabs out and in per rank, each about
    ``1/world`` of a stack plus the stencil margin, instead of ``world - 1`` whole stacks with a broadcast.
    """
    assert mine.layout == COLUMNS, "a band is a contiguous row range of a column stack"
    rank, world = _rank_world(comm)
    ranges, local_plan = _band_layout(plan, rank, world)
    lo, hi = ranges[rank]
    bands = [Stack.empty(hi - lo, mine.n_lev, mine.dtype, mine.device, COLUMNS) for _ in range(world)]
    if comm is not None:  # atx_exchange: one grouped send/recv, the own slab a device copy
        comm.exchange([mine.data[r_lo:r_hi] if r_hi > r_lo else None for r_lo, r_hi in ranges],
                      [b.data if hi > lo else None for b in bands])
        return bands, local_plan
    bands[rank].data.copy_(mine.data[lo:hi])
    # Host-side backends (gloo: rehearsals, CPU tests) get the slabs through pinned host memory.  Handed a device tensor, gloo's
    # point-to-point path lets its TCP transport read HBM through the PCIe BAR — uncached host reads of device memory — and
    # the band exchange of two O1280 stacks took 150 s against 1.6 s for a broadcast of twice the bytes, which gloo stages itself
    # (profiles/r02_bench_n2_rehearsal_shared_gpu.json).  RCCL (backend nccl) takes the device slabs as they are.
    staged = mine.data.is_cuda and dist.get_backend(_data_group) != "nccl"
    ops, landing = [], []
    for r in range(world):
        if r == rank:
            continue
        r_lo, r_hi = ranges[r]
        if r_hi > r_lo:
            out = mine.data[r_lo:r_hi]
            if staged:
                out = torch.empty(out.shape, dtype=out.dtype, pin_memory=True).copy_(out)
            ops.append(dist.P2POp(dist.isend, out, r, group=_data_group))
        if hi > lo:
            into = bands[r].data
            if staged:
                into = torch.empty(into.shape, dtype=into.dtype, pin_memory=True)
                landing.append((bands[r].data, into))
            ops.append(dist.P2POp(dist.irecv, into, r, group=_data_group))
    if ops:
        for work in dist.batch_isend_irecv(ops):
            work.wait()
    for device_side, host_side in landing:
        device_side.copy_(host_side, non_blocking=True)
    return bands, local_plan


def sharded_regrid(plan: GatherPlan, src: Stack, rank: int | None = None, world: int | None = None) -> Stack:
    """This rank's slice of ``plan`` applied to the (replicated) source stack."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return plan.shard(rank, world).apply(src)


def _streams(device):
    """(compute stream, a fresh side stream) — a seam the ordering test replaces with recording doubles."""
    return torch.cuda.current_stream(), torch.cuda.Stream(device=device)


def _on_stream(stream):
    return torch.cuda.stream(stream)


def _event():
    return torch.cuda.Event()


def _record_stream(tensor, stream) -> None:
    tensor.record_stream(stream)


def pipelined_sharded_regrid(plan: GatherPlan, mine: Stack, comm=None) -> list[Stack]:
    """End-to-end form of the target-sharded step: every rank contributes one source stack; the broadcast of stack
    ``r + 1`` runs (on the collective's own stream) while this rank interpolates its target slice of stack ``r``.
    Returns this rank's slice of every rank's stack, in rank order.  Two source buffers are alive at a time instead of
    ``world`` (SURVEY.md §7 "broadcast >> kernel": chunk by 137-level stack and double-buffer)."""
    rank, world = _rank_world(comm)
    local = plan.shard(rank, world)

    def incoming(r: int) -> Stack:
        return mine if r == rank else Stack.empty(mine.n_pts, mine.n_lev, mine.dtype, mine.device, mine.layout)

    outs: list[Stack] = []
    if comm is not None:
        # the same pipeline on explicit HIP streams: broadcasts are enqueued on a side stream, the launch of stack r waits
        # (on the device, via an event) for broadcast r only, so broadcast r + 1 runs under it.
        #
        # Ordering of the receive buffers.  Buffer r + 1 is ALLOCATED on the compute stream (torch's caching allocator) and first
        # WRITTEN by the broadcast on the side stream.  From the third stack on the allocator may hand back the block of stack
        # r - 1 — free for the side stream as soon as broadcast r - 1 is over, while `local.apply(r - 1)` may still be queued or
        # reading it on the compute stream.  So before EVERY broadcast the side stream waits for what the compute stream holds
        # at that moment (launches up to r - 1: broadcast r + 1 still runs under launch r, the overlap is kept).  torch's own nccl
        # path below gets the same guarantee from ProcessGroupNCCL, which makes its stream wait for the current stream per collective.
        compute, side = _streams(mine.device)

        def start(r: int):
            b = incoming(r)
            side.wait_stream(compute)  # `mine` being written, and every launch still reading a block `b` may now occupy
            with _on_stream(side):
                comm.bcast(b.data, r)
                done = _event()
                done.record(side)
            _record_stream(b.data, side)
            return b, done

        buf, done = start(0)
        for r in range(world):
            compute.wait_event(done)
            current = buf
            if r + 1 < world:
                buf, done = start(r + 1)
            outs.append(local.apply(current))
        return outs
    buf = incoming(0)
    work = dist.broadcast(buf.data, src=0, async_op=True, group=_data_group)
    for r in range(world):
        work.wait()  # the compute stream waits for broadcast r; the host does not block on the GPU
        current = buf
        if r + 1 < world:
            buf = incoming(r + 1)
            work = dist.broadcast(buf.data, src=r + 1, async_op=True, group=_data_group)  # overlaps with the launch below
        outs.append(local.apply(current))
    return outs


def gather_target_shards(local: Stack, plan: GatherPlan, comm=None) -> Stack:
    """All target slices of ``plan`` on every rank (column layout: each slice is a contiguous row range)."""
    assert local.layout == COLUMNS, "target shards are row ranges of a column stack"
    rank, world = _rank_world(comm)
    n_tgt = plan.n_tgt
    full = Stack.empty(n_tgt, local.n_lev, local.dtype, local.device, COLUMNS)
    if comm is not None:  # atx_gather_shards: the slices are byte ranges of the column stack
        bounds = plan.bounds(world)
        assert local.n_pts == bounds[rank + 1] - bounds[rank]
        full.data[bounds[rank]:bounds[rank + 1]].copy_(local.data)
        row_bytes = full.pitch * full.data.element_size()
        comm.gather_shards(full.data, [b * row_bytes for b in bounds])
        return full
    for r in range(world):
        lo, hi = plan.shard_range(r, world)
        if r == rank:
            assert local.n_pts == hi - lo
            full.data[lo:hi].copy_(local.data)
        if hi > lo:
            dist.broadcast(full.data[lo:hi], src=r, group=_data_group)
    return full
