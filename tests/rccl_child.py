"""Child process of tests/test_gpu_rccl.py — TEST INFRASTRUCTURE, not collected by pytest.

Runs the multi-GPU source exchange of anemoi_transform_amd.distributed on REAL RCCL with the ranks one MI355X allows
(world_size 1): communicator creation, collectives enqueued on HIP streams, the async-broadcast / compute-stream
ordering of the pipelined step, grouped send/recv, the shard gather — through torch.distributed's ``nccl`` backend
(``transport=torch``) or through the library's own C-ABI communicator ``atx_comm_*`` (``transport=atx``).  Every
result is compared with the oracle; a JSON verdict is printed on the last line.

    python tests/rccl_child.py torch|atx <port>
"""

from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    transport, port = sys.argv[1], sys.argv[2]
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as graft

    graft.load_package()
    from anemoi_transform_amd import distributed as atxd
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS, Stack
    from oracle import oracle  # checker only

    checks: dict[str, object] = {"transport": transport}
    comm = None
    if transport == "torch":
        # the first GPU call of this process is the communicator's: init with device_id, as bench.py does
        assert atxd.init_process_group("nccl") == (0, 1)
        checks["backend"] = dist.get_backend()
    else:
        torch.cuda.set_device(0)
        checks["rccl_version"] = native.Comm.rccl_version()
        comm = native.Comm(1, 0, native.Comm.unique_id())
        assert (comm.rank, comm.world) == (0, 1)
    dev = torch.device("cuda", 0)

    src_grid, tgt_grid = lookup("o48"), lookup([2.0, 2.0])
    n_src, n_tgt, n_lev = len(src_grid["latitudes"]), len(tgt_grid["latitudes"]), 37
    idx, w = interp.knn_inverse_distance(src_grid, tgt_grid, k=4)
    plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
    indptr = np.arange(n_tgt + 1) * 4
    rng = np.random.default_rng(11)
    host = (280.0 + rng.standard_normal((n_lev, n_src))).astype(np.float32)
    want = np.stack([oracle.csr_apply(w.astype(np.float32).reshape(-1), idx.reshape(-1), indptr, (n_tgt, n_src), f) for f in host])
    mine = Stack.from_fields(host, dev=dev)

    if comm is None:
        atxd.warm_up_transport()
    got = atxd.broadcast_stack(mine, 0, n_pts=n_src, n_lev=n_lev, dtype=mine.dtype, device=dev, comm=comm)
    checks["broadcast_stack"] = bool(np.array_equal(got.numpy(), host))
    stacks = atxd.exchange_stacks(mine, comm=comm)
    checks["exchange_stacks"] = len(stacks) == 1 and bool(np.array_equal(stacks[0].numpy(), host))
    gathered = atxd.exchange_stacks(mine, comm=comm, collective="all_gather")  # all_gather_into_tensor / atx_all_gather on real RCCL
    checks["all_gather"] = len(gathered) == 1 and bool(np.array_equal(gathered[0].numpy(), host))
    bands, local_plan = atxd.exchange_source_bands(mine, plan, comm=comm)
    checks["exchange_source_bands"] = bool(np.array_equal(local_plan.apply(bands[0]).numpy(), want))
    piped = atxd.pipelined_sharded_regrid(plan, mine, comm=comm)
    checks["pipelined_sharded_regrid"] = len(piped) == 1 and bool(np.array_equal(piped[0].numpy(), want))
    # the same ten times in a row on fresh buffers: an ordering bug between the collective's stream and the compute stream
    # would show as a stale read
    ok = True
    for rep in range(10):
        again = Stack.from_fields(host + np.float32(rep), dev=dev)
        out = atxd.pipelined_sharded_regrid(plan, again, comm=comm)[0]
        ref = plan.apply(again)
        ok = ok and bool(torch.equal(out.data, ref.data))
    checks["pipelined_repeat_10"] = ok
    full = atxd.gather_target_shards(piped[0], plan, comm=comm)
    checks["gather_target_shards"] = bool(np.array_equal(full.numpy(), want))

    # point-to-point on RCCL: a send/recv pair with this rank as its own peer, grouped (what the band exchange issues per peer)
    a = torch.arange(1 << 20, dtype=torch.float32, device=dev)
    b = torch.zeros_like(a)
    if comm is None:
        for work in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)]):
            work.wait()
        # an async broadcast followed by a kernel on the compute stream that reads the buffer
        c = torch.full((1 << 22,), 3.0, device=dev)
        work = dist.broadcast(c, src=0, async_op=True)
        work.wait()
        checks["async_broadcast_then_kernel"] = float((c * 2.0).sum().item()) == 6.0 * (1 << 22)
    else:
        # atx_exchange with a foreign peer entry is impossible at world 1; the own-slab path is a device copy
        comm.exchange([a], [b])
    torch.cuda.synchronize()
    checks["p2p_self"] = bool(torch.equal(a, b))

    if comm is None:
        dist.barrier()
        dist.destroy_process_group()
    else:
        comm.destroy()
    checks["ok"] = all(v for k, v in checks.items() if isinstance(v, bool))
    print(json.dumps(checks))


if __name__ == "__main__":
    main()
