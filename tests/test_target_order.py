"""Ordered traversal of the targets (`GatherPlan.order_targets`, `atx_regrid_ell_ordered`): the device visits the targets in column
blocks of the output grid; results are the natural order's, bit for bit — on the CPU double here, on the kernels in the GPU tests."""

from __future__ import annotations

import numpy as np
import pytest
import torch

import native_double
from anemoi_transform_amd import gather, interp, native
from anemoi_transform_amd.gather import GatherPlan, column_block_order
from anemoi_transform_amd.grids import lookup
from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack


def test_column_block_order_is_a_banded_permutation(monkeypatch):
    g = lookup([1.0, 1.0])  # 181 x 360
    assert column_block_order(g["latitudes"], g["longitudes"]) is None  # small grids keep their order
    monkeypatch.setattr(gather, "ORDER_MIN_TARGETS", 1000)
    order = column_block_order(g["latitudes"], g["longitudes"], block_points=60)
    n = len(g["latitudes"])
    assert order.dtype == np.int32 and np.array_equal(np.sort(order), np.arange(n))
    lon = g["longitudes"][order]
    band = np.floor(lon / 45.0).astype(int)
    assert np.all(np.diff(band) >= 0) and band.max() == 7  # asked for six, got EIGHT bands of 45 degrees, one after the other: whole bands per XCD
    first = order[band == 0]
    assert np.all(np.diff(first) > 0)  # inside a band: the grid's own (row-major) order
    assert column_block_order(g["latitudes"], np.zeros(n)) is None and column_block_order(g["latitudes"], g["longitudes"], block_points=100000) is None
    for asked, bands in ((180, 4), (120, 4), (90, 4), (72, 4), (45, 8), (36, 8), (30, 12), (20, 16)):  # the number of bands is a multiple of 4
        lon = g["longitudes"][column_block_order(g["latitudes"], g["longitudes"], block_points=asked)]
        assert len(np.unique(np.floor(lon / (360.0 / bands)).astype(int)[np.r_[True, np.diff(np.floor(lon / (360.0 / bands)).astype(int)) != 0]])) == bands, asked


@pytest.mark.parametrize("k", [1, 4])
def test_ordered_plan_equals_natural_order_on_the_double(monkeypatch, k):
    native_double.install(monkeypatch)
    monkeypatch.setattr(gather, "ORDER_MIN_TARGETS", 100)
    src, tgt = lookup("o16"), lookup([10.0, 10.0])
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    idx, w = interp.knn_inverse_distance(src, tgt, k=k)
    x = Stack.from_fields(280.0 + np.random.default_rng(2).standard_normal((5, n_src)), dev=torch.device("cpu"))
    natural = GatherPlan(n_src, n_tgt, index=idx, weights=w if k > 1 else None)
    ordered = GatherPlan(n_src, n_tgt, index=idx, weights=w if k > 1 else None)
    order = column_block_order(tgt["latitudes"], tgt["longitudes"], block_points=8)
    assert order is not None
    ordered.order_targets(order)
    want = natural.apply(x).numpy()
    assert np.array_equal(ordered.apply(x).numpy(), want)
    assert all(np.array_equal(a.numpy(), want) for a in ordered.apply_many([x, x]))
    assert np.array_equal(ordered.apply(x.to_layout(FIELDS)).numpy(), want)  # field-major stacks ignore the order
    for world in (2, 3):
        for rank in range(world):
            part = ordered.shard(rank, world)
            lo, hi = ordered.shard_range(rank, world)
            assert part.order is not None and np.array_equal(np.sort(part.order), np.arange(hi - lo))
            assert np.array_equal(part.apply(x).numpy(), want[:, lo:hi])
    if k > 1:  # the same through a general CSR plan (ragged rows): the permuted CSR matrix + tgt_rows
        keep = (np.arange(idx.size) % 5 != 0).reshape(idx.shape)
        indptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))])
        csr_n = GatherPlan(n_src, n_tgt, csr=(w[keep], idx[keep], indptr))
        csr_o = GatherPlan(n_src, n_tgt, csr=(w[keep], idx[keep], indptr)).order_targets(order)
        ref = csr_n.apply(x).numpy()
        assert np.array_equal(csr_o.apply(x).numpy(), ref)
        lo, hi = csr_o.shard_range(1, 2)
        assert np.array_equal(csr_o.shard(1, 2).apply(x).numpy(), ref[:, lo:hi]) and csr_o.shard(1, 2).order is not None
    with pytest.raises(ValueError):
        ordered.order_targets(np.zeros(n_tgt, dtype=np.int64))
    ordered.order_targets(None)
    assert ordered.order is None and np.array_equal(ordered.apply(x).numpy(), want)


def test_regrid_filter_orders_large_targets(monkeypatch):
    """The interpolators of the `regrid` filter order their plans by the output grid; small grids are left alone."""
    native_double.install(monkeypatch)
    from anemoi_transform_amd.filters import create_filter_by_name

    monkeypatch.setattr(gather, "ORDER_MIN_TARGETS", 100)
    monkeypatch.setattr(gather, "ORDER_BLOCK_POINTS", 8)
    monkeypatch.setattr(gather, "ORDER_MIN_K", 1)  # (the policy orders long rows only; the mechanism is what is under test)
    from anemoi_transform_amd.fields import fieldlist_from_dicts

    src = lookup("o16")
    specs = [{"param": "t", "values": np.random.default_rng(1).standard_normal(len(src["latitudes"])), "latitudes": src["latitudes"],
              "longitudes": src["longitudes"], "valid_datetime": "2020-01-01T00:00:00Z"}]
    small = create_filter_by_name("regrid", in_grid="o16", out_grid=[10.0, 10.0], method="nearest")
    out = small.forward(fieldlist_from_dicts(specs))
    assert small.interpolator._plan.order is not None
    monkeypatch.setattr(gather, "ORDER_MIN_TARGETS", 10**9)
    plain = create_filter_by_name("regrid", in_grid="o16", out_grid=[10.0, 10.0], method="nearest")
    ref = plain.forward(fieldlist_from_dicts(specs))
    assert plain.interpolator._plan.order is None
    assert np.array_equal(out[0].to_numpy(flatten=True), ref[0].to_numpy(flatten=True))
    # the policy itself: k = 1 .. 4 keep the natural order whatever the grid
    monkeypatch.setattr(gather, "ORDER_MIN_TARGETS", 100)
    monkeypatch.setattr(gather, "ORDER_MIN_K", 5)
    g = lookup([10.0, 10.0])
    assert gather.target_order_for(g["latitudes"], g["longitudes"], 4) is None and gather.target_order_for(g["latitudes"], g["longitudes"], 8) is not None
    assert gather.target_order_for(g["latitudes"], g["longitudes"], None) is None
