"""Randomised differential test of the code BETWEEN the plugin API and the kernels: `Pipeline.forward` with its fusion
(`filters/fusion.py`: stage folding, metadata followed through renames, one program per stack, point masks windowed to shards),
`fields.group_into_stacks` (host lists regrouped by variable and cut at `MAX_STACK_LEVELS`, device stacks used in place, sparse
selections) and `RegridFilter(shard=)`, against the same filters run one after the other (`ATX_NO_FUSION=1`, the reference's own
loop: R: workflows/pipeline.py:46-48) on random FieldLists and random pipelines.

Seeded: every run draws the same cases.  What varies: the number of fields, their parameters and order (level by level, variable
by variable, shuffled), one or two source grids in one list, NaNs, host or device-resident inputs (the latter a re-ordered subset of
the levels of one stack), the head of the pipeline (none / nearest / k = 4 matrix / index mask, whole or one shard of 2-3), and a
tail of 1-6 filters drawn from the per-point family (fusable), a stream-mask `apply_mask` and `remove_nans` (which cut a fused
segment).  Held: same length, same order, same `param` / `levelist` / `units`, same values bit for bit (NaNs in the same places),
same grid points, same target window — and, for host float64 lists behind an unsharded head, the same fields as the ORACLE's
filter-level restatements of the reference (`oracle.filter_*`) chained the same way.  `ATX_FUSION_SEEDS=first:count` widens the sweep for a soak.
"""

from __future__ import annotations

import os

import numpy as np
import pytest

from anemoi_transform_amd import fields as fields_mod
from anemoi_transform_amd import interp
from anemoi_transform_amd.fields import FieldList
from anemoi_transform_amd.filters import create_filter_by_name
from anemoi_transform_amd.grids import lookup
from oracle import oracle

import native_double
from test_filters import test_source

_FIRST, _COUNT = (int(v) for v in os.environ.get("ATX_FUSION_SEEDS", "0:0").split(":"))
SEEDS = range(_FIRST, _FIRST + _COUNT) if _COUNT else range(40)
PARAMS = ["t", "q", "orog", "z", "lnsp", "u", "sd"]


@pytest.fixture(params=["double", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request, monkeypatch):
    if request.param == "double":
        native_double.install(monkeypatch)
    return request.param


def random_fields(rng, grids):
    """Specs of a FieldList: 1-40 fields over one or two grids, parameters in one of three list orders."""
    n = int(rng.choice([1, 2, 3, 5, 8, 13, 24, 40]))
    names = list(rng.choice(PARAMS, size=int(rng.integers(1, 5)), replace=False))
    order = rng.choice(["level by level", "variable by variable", "shuffled"])
    labels = [(names[i % len(names)], i // len(names) + 1) for i in range(n)]
    if order == "variable by variable":
        labels.sort(key=lambda pl: names.index(pl[0]))
    elif order == "shuffled":
        labels = [labels[i] for i in rng.permutation(n)]
    two_grids = len(grids) > 1 and rng.random() < 0.3
    specs = []
    for i, (name, level) in enumerate(labels):
        grid = grids[int(rng.integers(0, 2))] if two_grids else grids[0]
        m = len(grid["latitudes"])
        if name == "lnsp":
            values = np.log(9.0e4 + 2.0e4 * rng.random(m))
        elif name in ("orog", "z"):
            values = rng.uniform(-100.0, 6000.0, m)
        else:
            values = 250.0 + 60.0 * rng.random(m)
        if rng.random() < 0.4:
            values[rng.random(m) < 0.05] = np.nan
        dtype = np.float32 if rng.random() < 0.15 else np.float64
        specs.append({"param": name, "levelist": level, "values": values.astype(dtype), "latitudes": grid["latitudes"],
                      "longitudes": grid["longitudes"], "units": "K" if name == "t" else "1", "valid_datetime": "2020-01-01T00:00:00Z"})
    if len({s["values"].dtype for s in specs}) > 1 and rng.random() < 0.5:  # mostly one width per list
        for s in specs:
            s["values"] = s["values"].astype(np.float64)
    return specs


def random_head(rng, src, tgt, tables):
    """`None` or the config of a `regrid` head: nearest / matrix / mask, whole or one shard."""
    kind = rng.choice(["none", "none", "nearest", "matrix", "mask"])
    if kind == "none":
        return None
    shard = None
    if rng.random() < 0.4:
        world = int(rng.integers(2, 4))
        shard = (int(rng.integers(0, world)), world)
    if kind == "nearest":
        cfg = dict(in_grid=src, out_grid=tgt, method="nearest")
    elif kind == "matrix":
        cfg = dict(matrix=tables["matrix"])
    else:
        cfg = dict(mask=tables["index_mask"] if rng.random() < 0.5 else tables["bool_mask"])
    if shard is not None:
        cfg["shard"] = shard
    return cfg


def random_tail(rng, n_points_for_mask, tmp_path, seed, present):
    """1-6 filter configs; most fusable, some that cut the fused segment."""
    pool = [
        lambda: ("rescale", dict(scale=float(rng.choice([1.0, 2.0, 0.5, 1.8])), offset=float(rng.choice([0.0, -273.15, 32.0])), param=str(rng.choice(PARAMS)))),
        lambda: ("convert", dict(unit_in="K", unit_out="degC", param="t")),
        lambda: ("orog_to_z", {}),
        lambda: ("orog_to_z_fields", {}),
        lambda: ("z_to_orog", {}),
        lambda: ("clip", dict(param=str(rng.choice(PARAMS)), minimum=float(rng.uniform(200.0, 270.0)), **({"maximum": float(rng.uniform(280.0, 320.0))} if rng.random() < 0.5 else {}))),
        lambda: ("impute_nans", dict(param=[str(p) for p in rng.choice(PARAMS, size=2, replace=False)], value=float(rng.choice([0.0, -1.0])))),
        lambda: ("lnsp_to_sp", {}),
        lambda: ("rename", dict(param={"q": "qq"}) if rng.random() < 0.5 else dict(param={"t": "temp"})),
        lambda: ("apply_mask_file", None),
        lambda: ("remove_nans", {} if rng.random() < 0.6 else dict(param=str(rng.choice(PARAMS)))),
        lambda: ("noop", {}),
        # the mask taken from a field of the stream (R: apply_mask.py:194-218): first field of that name, consumed unless `return_mask`
        lambda: ("apply_mask", dict(mask_param=str(rng.choice(present if rng.random() < 0.85 else PARAMS)),
                                    **(dict(mask_value=float(rng.choice([0.0, 1.0, 250.0]))) if rng.random() < 0.3 else
                                       dict(threshold=float(rng.uniform(240.0, 320.0)), threshold_operator=str(rng.choice([">", "<", ">=", "<=", "gt", "le", "ne"])))),
                                    **({"param": [str(p) for p in rng.choice(PARAMS, size=2, replace=False)]} if rng.random() < 0.6 else {}),
                                    **({"rename": "m"} if rng.random() < 0.3 else {}), return_mask=bool(rng.random() < 0.4))),
    ]
    weights = np.array([3, 2, 2, 1, 1, 2, 2, 1, 1, 2, 0.5, 0.5, 1.5])
    tail = []
    for _ in range(int(rng.integers(1, 7))):
        name, cfg = pool[int(rng.choice(len(pool), p=weights / weights.sum()))]()
        if name == "apply_mask_file":
            if n_points_for_mask is None:
                continue
            path = str(tmp_path / f"mask-{seed}-{len(tail)}.npy")
            np.save(path, (rng.random(n_points_for_mask) < 0.3).astype(np.float64))
            selected = [str(p) for p in rng.choice(PARAMS, size=int(rng.integers(1, 4)), replace=False)]
            cfg = dict(path=path, mask_value=1, param=selected, **({"rename": "masked"} if rng.random() < 0.3 else {}))
            name = "apply_mask"
        tail.append((name, cfg))
    return tail or [("noop", {})]


def build(head, tail):
    filters = ([create_filter_by_name("regrid", **head)] if head is not None else []) + [create_filter_by_name(n, **c) for n, c in tail]
    pipeline = filters[0]
    for f in filters[1:]:
        pipeline = pipeline | f
    return pipeline


def run(pipeline, data):
    out = pipeline.forward(data)
    return list(out)


def describe(f):
    lat, lon = f.grid_points()
    return (f.metadata("param"), f.metadata("levelist", default=None), f.metadata("units", default=None), tuple(np.shape(lat)),
            f.target_range() if hasattr(f, "target_range") else None)


def oracle_chain(specs, head, tail, src, tgt, tables):
    """The same pipeline on the oracle's filter-level restatements of the reference (oracle/oracle.py `filter_*`), or None when a
    stage has none (rename) or the head is sharded."""
    fields = [dict(s) for s in specs]
    if head is not None:
        if "shard" in head:
            return None
        if "method" in head:
            fields = oracle.filter_regrid_nearest(fields, in_grid=src, out_grid=tgt)
        elif "matrix" in head:
            fields = oracle.filter_regrid_matrix(fields, matrix=tables["matrix"])
        else:
            fields = oracle.filter_regrid_mask(fields, mask=head["mask"])
    for name, cfg in tail:
        if name == "rescale":
            fields = oracle.filter_rescale(fields, **cfg)
        elif name == "convert":
            fields = [dict(f, units="degC") if f.get("param") == "t" else f for f in oracle.filter_rescale(fields, scale=1.0, offset=-273.15, param="t")]
        elif name in ("orog_to_z", "orog_to_z_fields"):
            fields = oracle.filter_orog_to_z(fields)
        elif name == "z_to_orog":
            fields = oracle.filter_orog_to_z(fields, backward=True)
        elif name == "clip":
            fields = oracle.filter_clip(fields, **cfg)
        elif name == "impute_nans":
            fields = oracle.filter_impute_nans(fields, **cfg)
        elif name == "lnsp_to_sp":
            fields = oracle.filter_lnsp_to_sp(fields)
        elif name == "apply_mask" and "path" in cfg:
            fields = oracle.filter_apply_mask(fields, mask_values=np.load(cfg["path"]), mask_value=cfg["mask_value"], param=cfg["param"],
                                              rename=cfg.get("rename"))
        elif name == "apply_mask":
            fields = oracle.filter_apply_mask(fields, **cfg)
        elif name == "remove_nans":
            fields = oracle.filter_remove_nans(fields, **cfg)
        elif name == "noop":
            pass
        else:
            return None
    return fields


@pytest.mark.parametrize("seed", SEEDS)
def test_fused_pipeline_equals_filter_by_filter(engine, seed, tmp_path, monkeypatch):
    rng = np.random.default_rng(90_000 + seed)
    src, other, tgt = lookup("o16"), lookup("o8"), lookup([20.0, 20.0])
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    idx, w = interp.knn_inverse_distance(src, tgt, k=4)
    keep = np.sort(rng.choice(n_src, size=n_src // 3, replace=False))
    tables = {"matrix": {**interp.ell_to_csr(idx, w, n_src), "in_latitudes": src["latitudes"], "in_longitudes": src["longitudes"],
                         "out_latitudes": tgt["latitudes"], "out_longitudes": tgt["longitudes"]},
              "index_mask": keep, "bool_mask": np.isin(np.arange(n_src), keep)}
    head = random_head(rng, src, tgt, tables)
    # a head needs ONE source grid; without a head the list may mix two
    specs = random_fields(rng, [src] if head is not None else [src, other])
    # where a file mask can apply: the points the fields have when the mask stage runs (a shard knows its window of the full grid)
    if head is None:
        sizes = {s["values"].size for s in specs}
        n_mask = sizes.pop() if len(sizes) == 1 else None
    elif "mask" in head:
        n_mask = None if head.get("shard") else len(keep)
    else:
        n_mask = n_tgt
    tail = random_tail(rng, n_mask, tmp_path, seed, sorted({s["param"] for s in specs}))
    if rng.random() < 0.15:
        monkeypatch.setattr(fields_mod, "MAX_STACK_LEVELS", 4)  # long lists in several stacks

    data = test_source(specs).ds
    device_input = head is not None and rng.random() < 0.35
    if device_input:
        # device-resident input: the fields become levels of ONE stack (a first nearest-neighbour regrid onto their own grid), then a
        # re-ordered subset of them is handed on
        on_device = list(create_filter_by_name("regrid", in_grid=src, out_grid=src, method="nearest").forward(data))
        pick = rng.permutation(len(on_device))[: max(1, int(len(on_device) * rng.uniform(0.4, 1.0)))]
        data = FieldList([on_device[i] for i in pick])

    def outcome(no_fusion):
        if no_fusion:
            monkeypatch.setenv("ATX_NO_FUSION", "1")
        else:
            monkeypatch.delenv("ATX_NO_FUSION", raising=False)
        try:
            return run(build(head, tail), data), None
        except Exception as e:  # noqa: BLE001 - an invalid combination must be invalid BOTH ways, with the same exception type
            return None, e

    fused, fused_error = outcome(False)
    plain, plain_error = outcome(True)
    what = (seed, head if head is None else {k: (v if k in ("method", "shard") else "...") for k, v in head.items()}, tail)
    assert (fused_error is None) == (plain_error is None), (what, fused_error, plain_error)
    if os.environ.get("ATX_FUSION_STATS"):  # soak aid: what the seeds turned out to be
        with open(os.environ["ATX_FUSION_STATS"], "a") as f:
            f.write(f"{seed}\t{engine}\t{'error ' + type(fused_error).__name__ + ': ' + str(fused_error)[:80] if fused_error else 'ok'}\t"
                    f"{len(specs)} fields\thead={None if head is None else sorted(head)}\ttail={[n for n, _ in tail]}\n")
    if fused_error is not None:
        assert type(fused_error) is type(plain_error), (what, fused_error, plain_error)
        return
    assert len(fused) == len(plain), what
    for i, (a, b) in enumerate(zip(fused, plain)):
        assert describe(a) == describe(b), (what, i)
        va, vb = a.to_numpy(flatten=True), b.to_numpy(flatten=True)
        assert va.dtype == vb.dtype and np.array_equal(va, vb, equal_nan=True), (what, i, a.metadata("param"))
        assert np.array_equal(np.signbit(va), np.signbit(vb)), (what, i)
        assert np.array_equal(a.grid_points()[0], b.grid_points()[0]) and np.array_equal(a.grid_points()[1], b.grid_points()[1]), (what, i)

    # Pipeline.backward (R: workflows/pipeline.py:50-64: the filters' backward transforms in reverse order), fused the same way
    reversible = {"rescale", "convert", "orog_to_z", "orog_to_z_fields", "z_to_orog", "lnsp_to_sp", "noop"}
    if head is None and all(n in reversible for n, _ in tail):
        def back(no_fusion):
            if no_fusion:
                monkeypatch.setenv("ATX_NO_FUSION", "1")
            else:
                monkeypatch.delenv("ATX_NO_FUSION", raising=False)
            try:
                return list(build(None, tail).backward(data)), None
            except Exception as e:  # noqa: BLE001
                return None, e

        (fused_b, err_f), (plain_b, err_p) = back(False), back(True)
        assert type(err_f) is type(err_p), (what, err_f, err_p)
        if err_f is None:
            assert len(fused_b) == len(plain_b), what
            for i, (a, b) in enumerate(zip(fused_b, plain_b)):
                assert describe(a) == describe(b), (what, "backward", i)
                assert np.array_equal(a.to_numpy(flatten=True), b.to_numpy(flatten=True), equal_nan=True), (what, "backward", i)
        monkeypatch.delenv("ATX_NO_FUSION", raising=False)

    # third leg: the oracle's restatement of the reference's filters, chained the same way (host float64 lists, unsharded heads)
    if device_input or any(s["values"].dtype != np.float64 for s in specs):
        return
    try:
        want = oracle_chain(specs, head, tail, src, tgt, tables)
    except Exception as e:  # noqa: BLE001
        raise AssertionError(f"the oracle refuses what the engine accepted: {type(e).__name__}: {e}; {what}") from e
    if want is None:
        return
    assert len(want) == len(fused), what
    for i, (a, w_) in enumerate(zip(fused, want)):
        assert a.metadata("param") == w_["param"], (what, i)
        got, ref = a.to_numpy(flatten=True), np.asarray(w_["values"]).ravel()
        assert got.shape == ref.shape, (what, i)
        if any(n == "lnsp_to_sp" for n, _ in tail) and w_["param"] == "sp":
            with np.errstate(all="ignore"):  # exp: <= 1 ulp from numpy's (tests/test_gpu_kernels.py), and whatever follows it in the chain
                close = np.isclose(got, ref, rtol=1e-14, atol=0.0, equal_nan=True) | (np.isinf(got) & np.isinf(ref))
            assert close.all(), (what, i)
        else:
            assert np.array_equal(got, ref, equal_nan=True), (what, i, w_["param"])
        assert np.array_equal(a.grid_points()[0], np.asarray(w_["latitudes"])) and np.array_equal(a.grid_points()[1], np.asarray(w_["longitudes"])), (what, i)
