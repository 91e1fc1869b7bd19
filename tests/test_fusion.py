"""Fused pipelines give exactly the results of the filter-by-filter pipeline (and of the oracle)."""

from __future__ import annotations

import numpy as np
import pytest

from anemoi_transform_amd import interp, native
from anemoi_transform_amd.filters import create_filter_by_name
from anemoi_transform_amd.filters import fusion
from anemoi_transform_amd.grids import lookup
from oracle import oracle

import native_double
from test_filters import collect_fields_by_param, synthetic_fields, test_source


@pytest.fixture(params=["double", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request, monkeypatch):
    if request.param == "double":
        native_double.install(monkeypatch)
    return request.param


@pytest.fixture
def launches(monkeypatch):
    """Counts kernel launches by wrapping the native entry points."""
    counts = {"regrid_ell": 0, "regrid_csr": 0, "pointwise_stack": 0}
    for name in counts:
        real = getattr(native, name)

        def wrapped(*a, _real=real, _name=name, **k):
            counts[_name] += 1
            return _real(*a, **k)

        monkeypatch.setattr(native, name, wrapped)
    return counts


def config5_pipeline(matrix, mask_path=None):
    filters = [
        create_filter_by_name("regrid", matrix=matrix),
        create_filter_by_name("orog_to_z"),
        create_filter_by_name("convert", unit_in="K", unit_out="degC", param="t"),
        create_filter_by_name("clip", param="q", minimum=270.0),
    ]
    if mask_path:
        filters.append(create_filter_by_name("apply_mask", path=mask_path, mask_value=1, param=["t", "z"], rename="masked"))
    return filters


def setup_case(tmp_path, ragged=False):
    src, tgt = lookup("o32"), lookup([5.0, 5.0])
    idx, w = interp.knn_inverse_distance(src, tgt, k=4)
    if ragged:
        keep = (np.arange(idx.size) % 5 != 0).reshape(idx.shape)
        matrix = dict(matrix_data=w[keep], matrix_indices=idx[keep].astype(np.int32),
                      matrix_indptr=np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.int32),
                      matrix_shape=np.array([len(idx), len(src["latitudes"])]))
    else:
        matrix = interp.ell_to_csr(idx, w, len(src["latitudes"]))
    matrix = {**matrix, "out_latitudes": tgt["latitudes"], "out_longitudes": tgt["longitudes"]}
    specs = synthetic_fields(src, 6, nan_frac=0.01)
    for i, p in enumerate(["t", "orog", "q", "t", "lsm", "q"]):
        specs[i]["param"] = p
    rng = np.random.default_rng(4)
    mask = (rng.random(len(tgt["latitudes"])) < 0.3).astype(float)
    mask_path = str(tmp_path / "tmask.npy")
    np.save(mask_path, mask)
    return specs, matrix, mask_path, mask.astype(bool)


@pytest.mark.parametrize("ragged", [False, True])
def test_regrid_chain_fused_equals_unfused_and_oracle(engine, launches, tmp_path, monkeypatch, ragged):
    specs, matrix, mask_path, mask = setup_case(tmp_path, ragged)
    source = test_source(specs)

    pipeline = source
    for f in config5_pipeline(matrix, mask_path):
        pipeline = pipeline | f
    fused = list(pipeline)
    assert launches["regrid_ell"] + launches["regrid_csr"] == 1 and launches["pointwise_stack"] == 0  # ONE launch

    monkeypatch.setenv("ATX_NO_FUSION", "1")
    pipeline = source
    for f in config5_pipeline(matrix, mask_path):
        pipeline = pipeline | f
    unfused = list(pipeline)
    assert launches["pointwise_stack"] >= 3

    base = oracle.filter_regrid_matrix([dict(s) for s in specs], matrix=matrix)
    want = oracle.filter_orog_to_z(base)
    want = oracle.filter_rescale(want, scale=1.0, offset=-273.15, param="t")
    want = oracle.filter_clip(want, param="q", minimum=270.0)
    want = oracle.filter_apply_mask(want, mask_values=mask.astype(float), mask_value=1, param=["t", "z"], rename="masked")

    assert [f.metadata("param") for f in fused] == [w["param"] for w in want] == [f.metadata("param") for f in unfused]
    assert [f.metadata("param") for f in fused] == ["t_masked", "z_masked", "q", "t_masked", "lsm", "q"]
    for a, b, w in zip(fused, unfused, want):
        assert np.array_equal(a.to_numpy(flatten=True), b.to_numpy(flatten=True), equal_nan=True)
        assert np.array_equal(a.to_numpy(flatten=True), np.asarray(w["values"]).ravel(), equal_nan=True)
        assert np.array_equal(a.grid_points()[0], matrix["out_latitudes"])
        assert a.metadata("levelist") == b.metadata("levelist")
    assert fused[0].metadata("units") == "degC"


@pytest.mark.parametrize("fuse", [True, False])
def test_target_sharded_pipeline_with_a_full_grid_mask(engine, tmp_path, monkeypatch, fuse):
    """Every rank of a target-sharded job runs `regrid(shard=(r, world)) | ... | apply_mask(path=<full-grid mask>)` on
    its own slice: the slices know which target points they hold and the mask is windowed accordingly; the slices
    concatenate to the unsharded result (fused and filter by filter)."""
    specs, matrix, mask_path, mask = setup_case(tmp_path)
    if not fuse:
        monkeypatch.setenv("ATX_NO_FUSION", "1")

    def run(shard):
        pipeline = test_source(specs)
        filters = config5_pipeline(matrix, mask_path)
        filters[0] = create_filter_by_name("regrid", matrix=matrix, shard=shard)
        for f in filters:
            pipeline = pipeline | f
        return list(pipeline)

    whole = run(None)
    world = 3
    parts = [run((r, world)) for r in range(world)]
    n_tgt = len(matrix["out_latitudes"])
    for i, w in enumerate(whole):
        assert w.target_range() is None
        ranges = [p[i].target_range() for p in parts]
        assert ranges[0][0] == 0 and ranges[-1][1] == n_tgt and all(r[2] == n_tgt for r in ranges)
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        joined = np.concatenate([p[i].to_numpy(flatten=True) for p in parts])
        assert np.array_equal(joined, w.to_numpy(flatten=True), equal_nan=True)
        assert np.array_equal(np.concatenate([p[i].grid_points()[0] for p in parts]), w.grid_points()[0])
        assert [p[i].metadata("param") for p in parts] == [w.metadata("param")] * world
    # a mask that is neither the slice's nor the whole grid's length is still an error
    np.save(str(tmp_path / "short.npy"), np.zeros(n_tgt - 1))
    bad = create_filter_by_name("apply_mask", path=str(tmp_path / "short.npy"), mask_value=1)
    with pytest.raises(IndexError):
        list(test_source(specs) | create_filter_by_name("regrid", matrix=matrix, shard=(0, world)) | bad)


def test_pointwise_run_without_regrid_is_one_launch(engine, launches):
    src = lookup("o16")
    specs = synthetic_fields(src, 4)
    for i, p in enumerate(["orog", "t", "q", "z"]):
        specs[i]["param"] = p
    source = test_source(specs)
    pipeline = (source | create_filter_by_name("orog_to_z") | create_filter_by_name("rescale", scale=2.0, offset=1.0, param="z")
                | create_filter_by_name("z_to_orog"))
    out = list(pipeline)
    assert launches["pointwise_stack"] == 1
    assert [f.metadata("param") for f in out] == ["orog", "t", "q", "orog"]
    x0, x3 = specs[0]["values"], specs[3]["values"]
    assert np.array_equal(out[0].to_numpy(), oracle.z_to_orog(oracle.rescale_forward(oracle.orog_to_z(x0), 2.0, 1.0)))
    assert np.array_equal(out[3].to_numpy(), oracle.z_to_orog(oracle.rescale_forward(x3, 2.0, 1.0)))
    assert out[1] is source.ds[1] and out[2] is source.ds[2]  # untouched fields pass by identity


def test_pipeline_backward_is_fused_and_equals_filter_by_filter(engine, launches, monkeypatch):
    """Pipeline.backward (R: workflows/pipeline.py:50-64) runs the filters' backward transforms in reverse order — here as
    one launch, with the results of the filter-by-filter loop."""
    src = lookup("o16")
    specs = synthetic_fields(src, 4)
    for i, p in enumerate(["z", "t", "q", "t"]):
        specs[i]["param"] = p
    filters = [create_filter_by_name("orog_to_z"), create_filter_by_name("rescale", scale=1.8, offset=32.0, param="t")]
    pipeline = filters[0] | filters[1]
    from anemoi_transform_amd.fields import FieldList

    fields = FieldList(list(test_source(specs)))
    got = list(pipeline.backward(fields))
    assert launches["pointwise_stack"] == 1
    monkeypatch.setenv("ATX_NO_FUSION", "1")
    want = list(filters[0].backward(filters[1].backward(fields)))
    assert [f.metadata("param") for f in got] == ["orog", "t", "q", "t"] == [f.metadata("param") for f in want]
    for a, b, spec in zip(got, want, specs):
        assert np.array_equal(a.to_numpy(), b.to_numpy(), equal_nan=True)
    assert np.array_equal(got[1].to_numpy(flatten=True), oracle.rescale_backward(specs[1]["values"], 1.8, 32.0))
    assert np.array_equal(got[0].to_numpy(flatten=True), oracle.z_to_orog(specs[0]["values"]))
    assert got[2] is fields[2]  # untouched fields pass through by identity
    # forward then backward through the pipeline is the identity up to rounding
    monkeypatch.delenv("ATX_NO_FUSION")
    specs[0]["param"] = "orog"
    fields = FieldList(list(test_source(specs)))
    back = list(pipeline.backward(pipeline.forward(fields)))
    assert [f.metadata("param") for f in back] == ["orog", "t", "q", "t"]
    for f, spec in zip(back, specs):
        np.testing.assert_allclose(f.to_numpy(flatten=True), spec["values"], rtol=1e-12)


def test_unfusable_filters_cut_the_segment(engine, launches):
    src, tgt = lookup("o16"), lookup([10.0, 10.0])
    specs = synthetic_fields(src, 3, nan_frac=0.05)
    specs[1]["param"] = "lsm"
    specs[1]["values"] = (np.asarray(specs[1]["values"]) > 280).astype(float)
    source = test_source(specs)
    pipeline = (source | create_filter_by_name("regrid", in_grid="o16", out_grid=[10.0, 10.0], method="nearest")
                | create_filter_by_name("rescale", scale=1.0, offset=-273.15, param="t")
                | create_filter_by_name("apply_mask", mask_param="lsm", mask_value=0)  # mask from the stream: not fusable
                | create_filter_by_name("remove_nans"))
    out = list(pipeline)
    base = oracle.filter_regrid_nearest([dict(s) for s in specs], in_grid=src, out_grid=tgt)
    want = oracle.filter_remove_nans(oracle.filter_apply_mask(
        oracle.filter_rescale(base, scale=1.0, offset=-273.15, param="t"), mask_param="lsm", mask_value=0))
    assert len(out) == len(want) == 2
    for f, w in zip(out, want):
        assert np.array_equal(f.to_numpy(flatten=True), w["values"], equal_nan=True)
        assert np.array_equal(f.grid_points()[1], w["longitudes"])


def test_flatten_and_stage_detection():
    a = create_filter_by_name("orog_to_z")
    b = create_filter_by_name("rescale", scale=1.0, offset=0.0, param="t")
    c = create_filter_by_name("remove_nans")
    p = (a | b) | c
    assert fusion.flatten(p.filters) == [a, b, c]
    assert fusion.as_stage(a) is not None and fusion.as_stage(b) is not None and fusion.as_stage(c) is None
    assert fusion.as_stage(b.reverse()) is not None
    assert fusion.as_stage(create_filter_by_name("clip", param="t", minimum=0.0).reverse()) is None  # clip is not reversible
    assert fusion.as_stage(create_filter_by_name("apply_mask", mask_param="lsm", mask_value=0)) is None


def test_sparse_selection_on_a_device_stack_keeps_levels_in_place(engine):
    """A per-point filter that selects most (>= half) of the levels of an HBM stack transforms the stack as a
    whole (unselected levels untouched) instead of copying the selected levels out first."""
    src, tgt = lookup("o16"), lookup([20.0, 20.0])
    specs = synthetic_fields(src, 6, nan_frac=0.01)
    for i, p in enumerate(["t", "t", "q", "t", "t", "orog"]):
        specs[i]["param"] = p
    on_device = create_filter_by_name("regrid", out_grid=tgt, method="nearest").forward(test_source(specs).ds)
    stack = on_device[0].stack_ref()[0]
    assert all(f.stack_ref()[0] is stack for f in on_device)
    out = create_filter_by_name("rescale", scale=2.0, offset=-1.0, param="t").forward(on_device)
    new_stack = out[0].stack_ref()[0]
    assert new_stack is not stack and new_stack.n_lev == stack.n_lev
    for i, f in enumerate(out):
        if specs[i]["param"] == "t":
            assert f.stack_ref() == (new_stack, on_device[i].stack_ref()[1])  # same level of the new stack as of the old one: no compaction copy
            assert np.array_equal(f.to_numpy(), oracle.rescale_forward(on_device[i].to_numpy(), 2.0, -1.0), equal_nan=True)
        else:
            assert f is on_device[i]
    # a minority selection is compacted as before
    out = create_filter_by_name("orog_to_z").forward(on_device)
    assert out[5].stack_ref()[0].n_lev == 1 and out[0] is on_device[0]
    assert np.array_equal(out[5].to_numpy(), oracle.orog_to_z(on_device[5].to_numpy()), equal_nan=True)


def test_fused_mask_only_meets_the_fields_it_masks_two_grids(engine, tmp_path, monkeypatch):
    """a FieldList with two grids — `rescale` touches param t on grid A, `apply_mask(path=)` (a grid-B mask)
    touches param sd on grid B.  Filter by filter (and in the reference) the mask never meets grid A; the fused run must
    not measure it against grid A either, and must give the same fields."""
    grid_a, grid_b = lookup("o16"), lookup("o32")
    specs = synthetic_fields(grid_a, 2) + synthetic_fields(grid_b, 2, seed=1)
    for s in specs[2:]:
        s["param"] = "sd"
    mask_path = str(tmp_path / "mask_b.npy")
    rng = np.random.default_rng(4)
    mask_b = (rng.random(len(grid_b["latitudes"])) < 0.3).astype(np.float64)
    np.save(mask_path, mask_b)

    def pipeline():
        return [create_filter_by_name("rescale", scale=2.0, offset=1.0, param="t"),
                create_filter_by_name("apply_mask", path=mask_path, mask_value=1, param="sd")]

    def run():
        out = test_source(specs)
        for f in pipeline():
            out = out | f
        return list(out)

    fused = run()
    monkeypatch.setenv("ATX_NO_FUSION", "1")
    plain = run()
    assert len(fused) == len(plain) == 4
    for a, b in zip(fused, plain):
        assert a.metadata("param") == b.metadata("param")
        assert np.array_equal(a.to_numpy(), b.to_numpy(), equal_nan=True)
    assert np.array_equal(fused[0].to_numpy(), oracle.rescale_forward(specs[0]["values"], 2.0, 1.0))
    assert np.array_equal(np.isnan(fused[2].to_numpy()), mask_b.astype(bool))


def test_a_list_that_alternates_between_variables_becomes_runs_of_levels(engine):
    """A FieldList that comes level by level (t, q, t, q, ...) is uploaded with the fields of one variable next to each other, so a
    filter chain that treats the variables differently compiles to a program of RUNS of levels (the by-value routes of the kernels)
    instead of one that changes at every level — and the results still come back in the order of the list."""
    from anemoi_transform_amd.filters import fusion

    src, tgt = lookup("o16"), lookup([20.0, 20.0])
    specs = synthetic_fields(src, 12)
    for i, s_ in enumerate(specs):
        s_["param"] = ("t", "q", "orog")[i % 3]
        s_["levelist"] = i // 3
    regrid = create_filter_by_name("regrid", in_grid="o16", out_grid=tgt, method="nearest")
    convert = create_filter_by_name("convert", unit_in="K", unit_out="degC", param="t")
    orog = create_filter_by_name("orog_to_z_fields")
    on_device = regrid.forward(test_source(specs).ds)
    levels = [f.stack_ref()[1] for f in on_device]
    by_param = {p: sorted(l for l, s_ in zip(levels, specs) if s_["param"] == p) for p in ("t", "q", "orog")}
    assert all(v == list(range(v[0], v[0] + 4)) for v in by_param.values())  # four consecutive levels per variable
    out = list(test_source(specs) | regrid | convert | orog)
    want = oracle.filter_regrid_nearest([dict(s_) for s_ in specs], in_grid=src, out_grid=tgt)
    for f, w, s_ in zip(out, want, specs):
        v = np.asarray(w["values"]).ravel()
        v = oracle.rescale_forward(v, 1.0, -273.15) if s_["param"] == "t" else (oracle.orog_to_z(v) if s_["param"] == "orog" else v)
        assert f.metadata("levelist") == s_["levelist"] and f.metadata("param") == ("z" if s_["param"] == "orog" else s_["param"])
        assert np.array_equal(f.to_numpy(flatten=True), v, equal_nan=True)
    assert fusion is not None
