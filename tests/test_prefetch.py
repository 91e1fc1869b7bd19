"""`prefetch_to_device`: the next FieldList goes up while the current one is worked on; the consumer sees the same fields."""

from __future__ import annotations

import numpy as np
import pytest

import native_double
from anemoi_transform_amd.fields import fieldlist_from_dicts
from anemoi_transform_amd.filters import create_filter_by_name
from anemoi_transform_amd.grids import lookup
from anemoi_transform_amd.prefetch import prefetch_to_device, to_device


def _lists(n_lists, n_fields, seed=0):
    g = lookup("o16")
    rng = np.random.default_rng(seed)
    out = []
    for d in range(n_lists):
        out.append(fieldlist_from_dicts([{"param": "t", "levelist": l, "values": 250.0 + rng.standard_normal(len(g["latitudes"])) + d,
                                          "latitudes": g["latitudes"], "longitudes": g["longitudes"], "valid_datetime": f"2020-01-0{d + 1}T00:00:00Z"}
                                         for l in range(n_fields)]))
    return out


def test_to_device_keeps_fields_order_and_metadata(monkeypatch):
    native_double.install(monkeypatch)
    (fl,) = _lists(1, 5)
    dev = to_device(fl)
    assert len(dev) == 5 and all(f.stack_ref() is not None for f in dev)
    assert len({id(f.stack_ref()[0]) for f in dev}) == 1  # one grid, one stack
    for a, b in zip(fl, dev):
        assert np.array_equal(a.to_numpy(flatten=True), b.to_numpy(flatten=True))
        assert a.metadata("levelist") == b.metadata("levelist") and a.metadata("param") == b.metadata("param")
        assert np.array_equal(a.grid_points()[0], b.grid_points()[0])
    again = to_device(dev)  # already there: the same field objects
    assert all(x is y for x, y in zip(dev, again))


def test_prefetch_yields_every_list_in_order_and_filters_run_on_them(monkeypatch):
    native_double.install(monkeypatch)
    lists = _lists(4, 3)
    regrid = create_filter_by_name("regrid", in_grid="o16", out_grid=[10.0, 10.0], method="nearest")
    want = [[f.to_numpy(flatten=True) for f in regrid.forward(fl)] for fl in lists]
    got = []
    for dev_fl in prefetch_to_device(iter(lists), depth=2):
        assert all(f.stack_ref() is not None for f in dev_fl)
        got.append([f.to_numpy(flatten=True) for f in regrid.forward(dev_fl)])
    assert len(got) == 4 and all(np.array_equal(a, b) for x, y in zip(got, want) for a, b in zip(x, y))


def test_prefetch_passes_errors_on_and_stops_when_closed(monkeypatch):
    native_double.install(monkeypatch)
    lists = _lists(3, 2)

    def broken():
        yield lists[0]
        raise RuntimeError("decoder failed")

    it = prefetch_to_device(broken())
    assert len(next(it)) == 2
    with pytest.raises(RuntimeError, match="decoder failed"):
        next(it)
    seen = []

    def counting():
        for fl in lists:
            seen.append(1)
            yield fl

    it = prefetch_to_device(counting(), depth=1)
    next(it)
    it.close()  # the consumer walks away: the producer stops, nothing hangs
    assert len(seen) <= 3
    with pytest.raises(ValueError):
        next(prefetch_to_device(iter(lists), depth=0))


@pytest.mark.gpu
def test_prefetch_on_the_device_uploads_on_its_own_stream(dev):
    """On the MI355X the producer thread uploads under its OWN stream and hands the consumer an event: the default stream carries no
    wait marker of the producer's, the stacks are usable as soon as the consumer's stream has waited, results equal the host-fed path."""
    import torch

    g = lookup("o96")
    rng = np.random.default_rng(5)
    lists = []
    for d in range(4):  # 60 float32 fields of O96 per list: 9.7 MB, above the pinned-staging threshold
        lists.append(fieldlist_from_dicts([{"param": "t", "levelist": l, "values": (250.0 + rng.standard_normal(len(g["latitudes"])) + d).astype(np.float32),
                                            "latitudes": g["latitudes"], "longitudes": g["longitudes"], "valid_datetime": f"2020-01-0{d + 1}T00:00:00Z"}
                                           for l in range(60)]))
    regrid = create_filter_by_name("regrid", in_grid="o96", out_grid=[1.0, 1.0], method="nearest")
    want = [[f.to_numpy(flatten=True) for f in regrid.forward(fl)] for fl in lists]
    got, streams = [], set()
    for dev_fl in prefetch_to_device(iter(lists), depth=2):
        stack = dev_fl[0].stack_ref()[0]
        assert stack.data.is_cuda and stack.dtype == torch.float32
        streams.add(torch.cuda.current_stream().cuda_stream)
        got.append([f.to_numpy(flatten=True) for f in regrid.forward(dev_fl)])
    assert len(streams) == 1  # the consumer stayed on its own stream throughout
    assert len(got) == 4 and all(np.array_equal(a, b) for x, y in zip(got, want) for a, b in zip(x, y))


@pytest.mark.gpu
@pytest.mark.parametrize("np_dtype", [np.float32, np.float64])
def test_chunked_pinned_upload_and_download_keep_every_row(dev, monkeypatch, np_dtype):
    """The host -> HBM path of `Stack.from_fields` above its pinned-staging threshold: rows copied by threads into two pinned chunks
    that alternate under asynchronous DMAs, then one relayout — with the thresholds lowered so that a small stack takes many chunks
    (at their real values only multi-GB uploads do: tests/test_gpu_plugin_fullsize.py samples seven fields of such a list).  Every row,
    both layouts, odd sizes, source arrays of another width (cast on the way) and non-contiguous views; then `Stack.numpy()` back through
    its pinned download."""
    import torch

    from anemoi_transform_amd import stack as stack_mod
    from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack

    monkeypatch.setattr(stack_mod, "_PINNED_MIN_BYTES", 1 << 10)
    rng = np.random.default_rng(12)
    for n_lev, n_pts, stage_bytes in ((37, 1001, 8 << 10), (5, 4099, 16 << 10), (64, 257, 4 << 10), (3, 70001, 1 << 20)):
        monkeypatch.setattr(stack_mod, "_STAGE_BYTES", stage_bytes)
        rows = [(250.0 + 30.0 * rng.standard_normal(n_pts)).astype(np_dtype) for _ in range(n_lev)]
        rows[1 % n_lev] = rows[1 % n_lev].astype(np.float64 if np_dtype == np.float32 else np.float32)  # another width: cast while staging
        wide = rng.standard_normal(2 * n_pts).astype(np_dtype)
        rows[2 % n_lev] = wide[::2]  # a strided view
        want = np.stack([np.asarray(r).astype(np_dtype) for r in rows])
        tdtype = torch.float32 if np_dtype == np.float32 else torch.float64
        for layout in (COLUMNS, FIELDS):
            st = Stack.from_fields(rows, dtype=tdtype, dev=dev, layout=layout)
            assert (st.n_lev, st.n_pts, st.layout, st.dtype) == (n_lev, n_pts, layout, tdtype)
            got = st.numpy()
            assert got.dtype == np_dtype and np.array_equal(got, want), (n_lev, n_pts, layout)
            for level in (0, n_lev // 2, n_lev - 1):
                assert np.array_equal(st.level_numpy(level), want[level])
