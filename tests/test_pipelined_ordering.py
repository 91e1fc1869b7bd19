"""Stream ordering of the double-buffered sharded step on the C-ABI transport (`pipelined_sharded_regrid(comm=...)`).

Buffer r + 1 is allocated on the compute stream and first written by a broadcast on the side stream; from the third stack on
torch's caching allocator may hand back the block of stack r - 1 while the launch that reads it is still queued on the compute
stream.  The side stream must therefore wait for the compute stream before EVERY broadcast (round 2 waited once, before the
loop — invisible at world 1, the only world size one GPU box can run on RCCL).  This test replays the step at world 4 with
recording doubles for the streams and the communicator and checks the order of events; the kernels are the oracle-backed double.
"""

from __future__ import annotations

import contextlib

import numpy as np
import torch

import native_double
from anemoi_transform_amd import distributed as atxd
from anemoi_transform_amd import interp
from anemoi_transform_amd.gather import GatherPlan
from anemoi_transform_amd.grids import lookup
from anemoi_transform_amd.stack import Stack


class _Stream:
    def __init__(self, name, log):
        self.name, self.log = name, log

    def wait_stream(self, other):
        self.log.append(("wait_stream", self.name, other.name))

    def wait_event(self, event):
        self.log.append(("wait_event", self.name, event.tag))


class _Event:
    def __init__(self, log):
        self.log, self.tag = log, None

    def record(self, stream):
        self.tag = f"after {self.log[-1][0]} {self.log[-1][-1]}"
        self.log.append(("record", stream.name, self.tag))


class _Comm:
    """rank 1 of 4: `bcast(buf, r)` delivers rank r's stack (rank 1's own buffer is left alone, as RCCL leaves the root's)."""

    def __init__(self, log, stacks, current):
        self.rank, self.world, self.log, self.stacks, self.current = 1, 4, log, stacks, current

    def bcast(self, tensor, root):
        assert self.current[0] == "side", "a broadcast was enqueued on the compute stream"
        self.log.append(("bcast", root))
        if root != self.rank:
            tensor.copy_(self.stacks[root].data)


def test_side_stream_waits_for_compute_before_every_broadcast(monkeypatch):
    native_double.install(monkeypatch)
    log: list = []
    current = ["compute"]
    compute, side = _Stream("compute", log), _Stream("side", log)

    @contextlib.contextmanager
    def on_stream(stream):
        current[0] = stream.name
        try:
            yield
        finally:
            current[0] = "compute"

    monkeypatch.setattr(atxd, "_streams", lambda device: (compute, side))
    monkeypatch.setattr(atxd, "_on_stream", on_stream)
    monkeypatch.setattr(atxd, "_event", lambda: _Event(log))
    monkeypatch.setattr(atxd, "_record_stream", lambda tensor, stream: log.append(("record_stream", stream.name)))

    src_grid, tgt_grid = lookup("o16"), lookup([10.0, 10.0])
    n_src, n_tgt, n_lev = len(src_grid["latitudes"]), len(tgt_grid["latitudes"]), 3
    idx, w = interp.knn_inverse_distance(src_grid, tgt_grid, k=4)
    plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
    host = [280.0 + np.random.default_rng(10 + r).standard_normal((n_lev, n_src)) for r in range(4)]
    stacks = [Stack.from_fields(h, dev=torch.device("cpu")) for h in host]
    comm = _Comm(log, stacks, current)

    real_apply = GatherPlan.apply

    def logged_apply(self, stack, *a, **k):
        assert current[0] == "compute"
        log.append(("apply", sum(1 for e in log if e[0] == "apply")))
        return real_apply(self, stack, *a, **k)

    monkeypatch.setattr(GatherPlan, "apply", logged_apply)
    outs = atxd.pipelined_sharded_regrid(plan, stacks[1], comm=comm)

    # results: this rank's target slice of every rank's stack
    shard = plan.shard(1, 4)
    assert len(outs) == 4
    for r, got in enumerate(outs):
        assert np.array_equal(got.numpy(), real_apply(shard, stacks[r]).numpy())

    # order: every broadcast is preceded — after the previous broadcast — by "side waits for compute", and that wait comes after
    # every launch enqueued so far (so the block a new receive buffer may occupy is no longer being read)
    events = [e for e in log if e[0] in ("wait_stream", "bcast", "apply", "wait_event")]
    assert [e for e in events if e[0] == "bcast"] == [("bcast", r) for r in range(4)]
    for r in range(4):
        at = events.index(("bcast", r))
        prev = events.index(("bcast", r - 1)) if r else -1
        waits = [i for i in range(prev + 1, at) if events[i] == ("wait_stream", "side", "compute")]
        assert waits, f"broadcast {r} was enqueued without the side stream waiting for the compute stream"
        launched_before_bcast = [e[1] for e in events[:at] if e[0] == "apply"]
        launched_before_wait = [e[1] for e in events[:waits[-1]] if e[0] == "apply"]
        assert launched_before_bcast == launched_before_wait == list(range(max(r - 1, 0)))  # launches up to r - 2 only: overlap kept
    # and the launch of stack r waits (on the device) for broadcast r, not for a later one
    for r in range(4):
        at = events.index(("apply", r))
        last_wait = [e for e in events[:at] if e[0] == "wait_event"][-1]
        assert last_wait == ("wait_event", "compute", f"after bcast {r}")
