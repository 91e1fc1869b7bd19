"""The N > 1 path with world_size 2 on the CPU (gloo): target-point sharding, one source
broadcast, optional gather of the shards.  Kernels are the oracle-backed double
(tests/native_double.py); what is under test is the sharding / communication logic that
bench.py and the driver's multi-GPU run rely on.
"""

from __future__ import annotations

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Patch:
    """Minimal monkeypatch stand-in for spawned workers."""

    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def _worker(rank: int, world: int, port: int, kind: str, tmpdir: str) -> None:
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as graft

    graft.load_package()
    import torch.distributed as dist

    import native_double
    from anemoi_transform_amd import distributed as atxd
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import Stack
    from oracle import oracle

    native_double.install(_Patch())
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    assert atxd.init_process_group("gloo") == (rank, world)

    src_grid, tgt_grid = lookup("o16"), lookup([10.0, 10.0])
    n_src, n_tgt, n_lev = len(src_grid["latitudes"]), len(tgt_grid["latitudes"]), 3
    idx, w = interp.knn_inverse_distance(src_grid, tgt_grid, k=4)
    if kind == "ell":
        plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
        indptr = np.arange(n_tgt + 1) * 4
        data, indices = w.reshape(-1), idx.reshape(-1)
    else:
        keep = (np.arange(idx.size) % 7 != 0).reshape(idx.shape)
        indptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))])
        data, indices = w[keep], idx[keep]
        plan = GatherPlan(n_src, n_tgt, csr=(data, indices, indptr))

    rng = np.random.default_rng(100 + rank)  # every rank owns a different source stack
    mine_host = 280.0 + rng.standard_normal((n_lev, n_src))
    mine = Stack.from_fields(mine_host, dev=torch.device("cpu"))

    # 1. sources exchanged once, then every rank interpolates its target slice of every stack
    atxd.warm_up_transport()  # what bench.py does before it times an exchange
    stacks = atxd.exchange_stacks(mine)
    assert len(stacks) == world
    for r, st in enumerate(stacks):
        expect = 280.0 + np.random.default_rng(100 + r).standard_normal((n_lev, n_src))
        assert np.array_equal(st.numpy(), expect)
        local = atxd.sharded_regrid(plan, st)
        full = atxd.gather_target_shards(local, plan)
        want = np.stack([oracle.csr_apply(data, indices, indptr, (n_tgt, n_src), f) for f in expect])
        assert np.array_equal(full.numpy(), want), f"rank {rank} stack {r}"

    # 1b. the same exchange as ONE all-gather: the same stacks, as views of one buffer
    gathered = atxd.exchange_stacks(mine, collective="all_gather")
    assert len(gathered) == world and all(np.array_equal(g.numpy(), s.numpy()) for g, s in zip(gathered, stacks))
    assert gathered[0].data.untyped_storage().data_ptr() == gathered[-1].data.untyped_storage().data_ptr()
    assert np.array_equal(atxd.sharded_regrid(plan, gathered[(rank + 1) % world]).numpy(), atxd.sharded_regrid(plan, stacks[(rank + 1) % world]).numpy())

    # 2. feeding a rank only the band of source columns its slice references gives the same rows
    shard = plan.shard(rank, world)
    lo, hi = atxd.source_band(shard)
    band = Stack.from_fields(mine_host[:, lo:hi], dev=torch.device("cpu"))
    got = atxd.rebase_plan(shard, lo, hi).apply(band).numpy()
    assert np.array_equal(got, shard.apply(mine).numpy())
    assert hi - lo < n_src  # a latitude band, not the whole grid

    # 3. band-limited exchange: every rank receives only its band of every other rank's stack (send / recv)
    bands, local_plan = atxd.exchange_source_bands(mine, plan)
    assert len(bands) == world and all(b.n_pts == hi - lo for b in bands)
    for r, band_stack in enumerate(bands):
        full_r = 280.0 + np.random.default_rng(100 + r).standard_normal((n_lev, n_src))
        assert np.array_equal(band_stack.numpy(), full_r[:, lo:hi])
        want = shard.apply(Stack.from_fields(full_r, dev=torch.device("cpu"))).numpy()
        assert np.array_equal(local_plan.apply(band_stack).numpy(), want)

    # 4. double-buffered form: broadcast of the next stack overlaps with the interpolation of the current one
    piped = atxd.pipelined_sharded_regrid(plan, mine)
    assert len(piped) == world
    for r, got_r in enumerate(piped):
        full_r = 280.0 + np.random.default_rng(100 + r).standard_normal((n_lev, n_src))
        assert np.array_equal(got_r.numpy(), shard.apply(Stack.from_fields(full_r, dev=torch.device("cpu"))).numpy())

    dist.barrier()
    with open(os.path.join(tmpdir, f"ok{rank}"), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,world", [("ell", 2), ("csr", 2), ("ell", 3), ("ell", 8)])
def test_target_sharded_regrid(tmp_path, kind, world):
    """World 2 for both plan kinds; world 3 as well: the double-buffered step then re-uses a receive buffer (stack r + 1 may get
    the block of stack r - 1) and every rank has two peers in the band exchange; world 8 — the node size north_star names: eight shards,
    seven peers per rank in the band exchange, eight broadcasts in the double-buffered step."""
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))



def _random_worker(rank: int, world: int, port: int, first_seed: int, n_cases: int, tmpdir: str) -> None:
    """Random matrices (tests/test_gather_random.py: uniform, ragged, long, empty rows, FEWER TARGETS THAN RANKS, no targets at all)
    through the sharded step in its three forms; every rank draws the same matrix from the shared seed and owns its own source stack."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as graft

    graft.load_package()
    import torch.distributed as dist
    from scipy.sparse import csr_array

    import native_double
    from anemoi_transform_amd import distributed as atxd
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.stack import Stack
    from test_gather_random import random_matrix

    native_double.install(_Patch())
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    assert atxd.init_process_group("gloo") == (rank, world)
    cpu = torch.device("cpu")
    for seed in range(first_seed, first_seed + n_cases):
        matrix, style = random_matrix(np.random.default_rng(70_000 + seed))
        n_tgt, n_src = (int(v) for v in matrix["matrix_shape"])
        n_lev = 1 + seed % 3
        m = csr_array((matrix["matrix_data"], matrix["matrix_indices"], matrix["matrix_indptr"]), shape=(n_tgt, n_src))
        plan = GatherPlan.from_matrix(matrix)
        sources = [280.0 + np.random.default_rng(1000 * seed + r).standard_normal((n_lev, n_src)) for r in range(world)]
        mine = Stack.from_fields(sources[rank], dev=cpu)
        lo, hi = plan.shard_range(rank, world)
        what = (seed, style, n_tgt, n_src, world, rank)

        def want(r):
            full = np.stack([m @ level for level in sources[r]])
            return full[:, lo:hi], full

        stacks = atxd.exchange_stacks(mine)
        for r, st in enumerate(stacks):
            local = atxd.sharded_regrid(plan, st)
            assert np.array_equal(local.numpy(), want(r)[0]), what
            assert np.array_equal(atxd.gather_target_shards(local, plan).numpy(), want(r)[1]), what
        bands, local_plan = atxd.exchange_source_bands(mine, plan)
        for r, band in enumerate(bands):
            assert np.array_equal(local_plan.apply(band).numpy(), want(r)[0]), what
        for r, got in enumerate(atxd.pipelined_sharded_regrid(plan, mine)):
            assert np.array_equal(got.numpy(), want(r)[0]), what
    dist.barrier()
    with open(os.path.join(tmpdir, f"ok{rank}"), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,first_seed", [(3, 0), (5, 40)])
def test_random_matrices_through_the_sharded_step(tmp_path, world, first_seed):
    """40 random matrices per world size — among them target grids smaller than the number of ranks (empty shards: zero-row tables,
    zero-byte bands) and none at all — through whole-stack exchange + sharded regrid + shard gather, the band-limited exchange, and the
    double-buffered step, each against scipy's `csr_array @ x` on the rank's window."""
    mp.spawn(_random_worker, args=(world, _free_port(), first_seed, 40, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
