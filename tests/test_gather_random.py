"""Randomised differential test of `GatherPlan` — the host logic between a regrid matrix and the gather kernels — against scipy's
own statement, `csr_array(...) @ x` (R: filters/fields/regrid.py:283-285,310).

`GatherPlan.from_matrix` decides, per matrix, between the fixed-k table, rows padded with index -1 (widths 9-11 / 13-15 widened to
12 / 16), and general CSR; `order_targets` permutes the device tables; `shard` cuts target slices balanced by traffic;
`distributed.source_band` / `rebase_plan` re-express a slice against a slab of the source; `apply_many` batches stacks; field-major
stacks with long rows detour through column stacks.  Whatever route a matrix takes, every level must be scipy's result bit for bit
(float64) — and bit for bit scipy's result in float32 for float32 stacks (the kernels keep scipy's summation order and the library is
built without contraction).

Seeded; matrices are drawn in the shapes the routes tell apart: uniform k (1 .. 20, 33, 64, 70), short ragged rows, rows of 9-16,
long ragged rows, empty rows, duplicated and unsorted column indices, zero weights, an empty target grid.
`ATX_GATHER_SEEDS=first:count` widens the sweep.
"""

from __future__ import annotations

import os

import numpy as np
import pytest
from scipy.sparse import csr_array

from anemoi_transform_amd.distributed import rebase_plan, source_band
from anemoi_transform_amd.gather import GatherPlan
from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack

import native_double

_FIRST, _COUNT = (int(v) for v in os.environ.get("ATX_GATHER_SEEDS", "0:0").split(":"))
SEEDS = range(_FIRST, _FIRST + _COUNT) if _COUNT else range(60)


@pytest.fixture(params=["double", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request, monkeypatch):
    if request.param == "double":
        native_double.install(monkeypatch)
        return None
    import torch

    return torch.device("cuda", 0)


def random_matrix(rng):
    n_src = int(rng.choice([1, 2, 7, 64, 257, 1000]))
    n_tgt = int(rng.choice([0, 1, 3, 63, 64, 65, 300, 1111]))
    style = rng.choice(["uniform", "uniform", "short ragged", "rows of 9-16", "long ragged", "anything"])
    if style == "uniform":
        lengths = np.full(n_tgt, int(rng.choice([1, 2, 3, 4, 5, 8, 9, 11, 12, 13, 16, 20, 33, 64, 70])))
    elif style == "short ragged":
        lengths = rng.integers(1, 9, n_tgt)
    elif style == "rows of 9-16":
        lengths = rng.integers(9, 17, n_tgt)
    elif style == "long ragged":
        lengths = rng.integers(0, 80, n_tgt)
    else:
        lengths = rng.integers(0, 24, n_tgt)
    if style != "uniform" and n_tgt and rng.random() < 0.5:
        lengths[rng.random(n_tgt) < 0.1] = 0  # empty rows: the statement's result there is 0.0
    indptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    nnz = int(indptr[-1])
    indices = rng.integers(0, n_src, nnz)  # duplicates within a row and unsorted columns included: scipy sums in storage order
    data = rng.normal(0.0, 1.0, nnz)
    if nnz and rng.random() < 0.3:
        data[rng.random(nnz) < 0.1] = 0.0
    if rng.random() < 0.5:  # interpolation-like rows: positive weights summing to one
        data = np.abs(data) + 1e-3
        row = np.repeat(np.arange(n_tgt), lengths)
        sums = np.bincount(row, weights=data, minlength=max(n_tgt, 1))[:n_tgt] if n_tgt else np.zeros(0)
        data = data / np.where(sums[row] > 0, sums[row], 1.0) if nnz else data
    return dict(matrix_data=data, matrix_indices=indices.astype(np.int32), matrix_indptr=indptr, matrix_shape=np.array([n_tgt, n_src])), style


def scipy_levels(matrix, x, np_dtype):
    n_tgt, n_src = (int(v) for v in matrix["matrix_shape"])
    m = csr_array((matrix["matrix_data"].astype(np_dtype), matrix["matrix_indices"], matrix["matrix_indptr"]), shape=(n_tgt, n_src))
    return np.stack([m @ level for level in x]) if len(x) else np.zeros((0, n_tgt), dtype=np_dtype)


def same_bits(got, want):
    assert got.dtype == want.dtype and got.shape == want.shape, (got.dtype, want.dtype, got.shape, want.shape)
    # (+0.0 and -0.0: an empty row is +0.0 in scipy; a sum of products can be -0.0 only if scipy's is)
    return np.array_equal(got, want, equal_nan=True) and np.array_equal(np.signbit(got), np.signbit(want))


@pytest.mark.parametrize("seed", SEEDS)
def test_every_route_of_a_gather_plan_gives_scipys_bits(engine, seed):
    rng = np.random.default_rng(70_000 + seed)
    matrix, style = random_matrix(rng)
    n_tgt, n_src = (int(v) for v in matrix["matrix_shape"])
    np_dtype = np.float64 if rng.random() < 0.6 else np.float32
    layout = COLUMNS if rng.random() < 0.7 else FIELDS
    n_lev = int(rng.choice([1, 2, 3, 4, 5, 9, 17]))
    x = (280.0 + 30.0 * rng.standard_normal((n_lev, n_src))).astype(np_dtype)
    if rng.random() < 0.3:
        x[rng.random(x.shape) < 0.02] = np.nan
    want = scipy_levels(matrix, x, np_dtype)
    what = (seed, style, n_tgt, n_src, np_dtype.__name__, "columns" if layout == COLUMNS else "fields", n_lev)

    plan = GatherPlan.from_matrix(matrix)
    assert (plan.n_tgt, plan.n_src) == (n_tgt, n_src)
    stack = Stack.from_fields(x, dev=engine, layout=layout)
    assert same_bits(plan.apply(stack).numpy(), want), what

    # the same matrix forced onto the general CSR route (from_matrix may have chosen a table)
    general = GatherPlan(n_src, n_tgt, csr=(matrix["matrix_data"], matrix["matrix_indices"], matrix["matrix_indptr"]))
    assert same_bits(general.apply(stack).numpy(), want), what

    # a visiting order changes nothing
    if n_tgt:
        order = rng.permutation(n_tgt)
        for p in (GatherPlan.from_matrix(matrix).order_targets(order), general.order_targets(order)):
            assert same_bits(p.apply(stack).numpy(), want), what
        general.order_targets(None)

    # target shards tile the grid and concatenate to the whole; a shard expressed against its own source band gives the same slice
    world = int(rng.integers(1, 6))
    bounds = plan.bounds(world)
    assert bounds[0] == 0 and bounds[-1] == n_tgt and all(a <= b for a, b in zip(bounds, bounds[1:])), (what, bounds)
    pieces = []
    for rank in range(world):
        shard = plan.shard(rank, world)
        assert plan.shard_range(rank, world) == (bounds[rank], bounds[rank + 1]) and shard.n_tgt == bounds[rank + 1] - bounds[rank]
        part = shard.apply(stack).numpy()
        pieces.append(part)
        lo, hi = source_band(shard)
        if hi > lo and layout == COLUMNS:
            slab = Stack(stack.data[lo:hi], hi - lo, stack.n_lev, COLUMNS)
            assert same_bits(rebase_plan(shard, lo, hi).apply(slab).numpy(), part), (what, rank)
    assert same_bits(np.concatenate(pieces, axis=1), want), what

    # several stacks of one shape in one batched call; a caller-kept output stack; a bound call
    others = [(x * np_dtype(1.0 + 0.25 * j)).astype(np_dtype) for j in range(int(rng.integers(1, 4)))]
    stacks = [stack] + [Stack.from_fields(o, dev=engine, layout=layout) for o in others]
    for got, src in zip(plan.apply_many(stacks), [x] + others):
        assert same_bits(got.numpy(), scipy_levels(matrix, src, np_dtype)), what
    keep = stack.new_like(n_pts=n_tgt)
    assert plan.apply(stack, out=keep) is keep and same_bits(keep.numpy(), want), what
    launch, out = plan.bind(stack)
    launch()
    assert same_bits(out.numpy(), want), what
