"""Child process of tests/test_gpu_rccl_stub.py — TEST INFRASTRUCTURE, not collected by pytest.

One rank of a world of 2-4 sharing the test box's single MI355X, with the C-ABI communicator ``atx_comm_*`` bound to the
host-staged stand-in of tests/rccl_stub (``ATX_RCCL_LIBRARY``): the multi-peer branches of ``atx_exchange`` (grouped send / recv
with per-peer byte counts), ``atx_gather_shards`` (one broadcast per peer range) and ``atx_bcast`` from every root, under the
functions of anemoi_transform_amd.distributed, with the real regrid kernels, against the oracle.

    python tests/rccl_stub_child.py <rank> <world> <token>
"""

from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    rank, world, token = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import numpy as np
    import torch

    import __graft_entry__ as graft

    graft.load_package()
    from anemoi_transform_amd import distributed as atxd
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import Stack
    from oracle import oracle  # checker only

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    checks: dict[str, object] = {"rank": rank, "rccl_version": native.Comm.rccl_version()}
    comm = native.Comm(world, rank, token.encode().ljust(native.COMM_ID_BYTES, b"\0"))

    src_grid, tgt_grid = lookup("o48"), lookup([2.0, 2.0])
    n_src, n_tgt, n_lev = len(src_grid["latitudes"]), len(tgt_grid["latitudes"]), 19
    idx, w = interp.knn_inverse_distance(src_grid, tgt_grid, k=4)
    plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
    indptr = np.arange(n_tgt + 1) * 4

    def stack_of(r: int, shift: float = 0.0) -> np.ndarray:  # every rank owns a different source stack, a pure function of its rank
        return 280.0 + np.random.default_rng(100 + r).standard_normal((n_lev, n_src)) + shift

    def regridded(host: np.ndarray) -> np.ndarray:
        return np.stack([oracle.csr_apply(w.reshape(-1), idx.reshape(-1), indptr, (n_tgt, n_src), f) for f in host])

    mine = Stack.from_fields(stack_of(rank), dev=dev)
    lo, hi = plan.shard_range(rank, world)

    # 1. every root broadcasts its stack (atx_bcast)
    stacks = atxd.exchange_stacks(mine, comm=comm)
    checks["exchange_stacks"] = len(stacks) == world and all(np.array_equal(st.numpy(), stack_of(r)) for r, st in enumerate(stacks))

    # 1b. the same exchange as ONE all-gather (atx_all_gather)
    gathered = atxd.exchange_stacks(mine, comm=comm, collective="all_gather")
    checks["all_gather"] = len(gathered) == world and all(np.array_equal(st.numpy(), stack_of(r)) for r, st in enumerate(gathered))

    # 2. band-limited exchange (atx_exchange: grouped send / recv, a different byte count per peer pair)
    bands, local_plan = atxd.exchange_source_bands(mine, plan, comm=comm)
    b_lo, b_hi = atxd.source_band(plan.shard(rank, world))
    ok = len(bands) == world and b_hi - b_lo < n_src
    for r, band in enumerate(bands):
        ok = ok and np.array_equal(band.numpy(), stack_of(r)[:, b_lo:b_hi])
        ok = ok and np.array_equal(local_plan.apply(band).numpy(), regridded(stack_of(r))[:, lo:hi])
    checks["exchange_source_bands"] = bool(ok)

    # 3. the double-buffered sharded step through the C-ABI transport, several times on fresh buffers (buffer r + 1 may reuse the
    #    block of stack r - 1 from the third stack on: the ordering fixed in round 3)
    ok = True
    for rep in range(4):
        again = Stack.from_fields(stack_of(rank, shift=float(rep)), dev=dev)
        outs = atxd.pipelined_sharded_regrid(plan, again, comm=comm)
        ok = ok and len(outs) == world
        for r, got in enumerate(outs):
            ok = ok and np.array_equal(got.numpy(), regridded(stack_of(r, shift=float(rep)))[:, lo:hi])
    checks["pipelined_sharded_regrid"] = bool(ok)

    # 4. the target slices of one stack assembled on every rank (atx_gather_shards: one broadcast per peer's byte range)
    local = plan.shard(rank, world).apply(stacks[1 % world])
    full = atxd.gather_target_shards(local, plan, comm=comm)
    checks["gather_target_shards"] = bool(np.array_equal(full.numpy(), regridded(stack_of(1 % world))))

    # 5. a raw exchange with unequal, partly empty slabs: rank r sends (r + 1) * (p + 1) * 1000 floats to peer p, nothing to peer r + 1
    send, recv = [], []
    for p in range(world):
        n_out = 0 if p == (rank + 1) % world and world > 1 else (rank + 1) * (p + 1) * 1000
        n_in = 0 if rank == (p + 1) % world and world > 1 else (p + 1) * (rank + 1) * 1000
        send.append(torch.full((n_out,), float(100 * rank + p), dtype=torch.float32, device=dev) if n_out else None)
        recv.append(torch.zeros(n_in, dtype=torch.float32, device=dev) if n_in else None)
    comm.exchange(send, recv)
    torch.cuda.synchronize()
    checks["raw_exchange"] = all(t is None or bool((t == float(100 * p + rank)).all()) for p, t in enumerate(recv))

    comm.destroy()
    checks["ok"] = all(v for v in checks.values() if isinstance(v, bool))
    print(json.dumps(checks))


if __name__ == "__main__":
    main()
