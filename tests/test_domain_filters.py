"""The reference's numpy-only domain filters — rodeo_opera_clipping, rodeo_opera_preprocessing, oras6_clipping, land_parameters
(anemoi_transform_amd/filters/domain.py) — written like the reference's tests where it has them
(R: tests/field_filters/test_rodeo_opera_clipping.py, test_rodeo_opera_preprocessing.py), against the oracle's restatement elsewhere.

Everything here is compare-and-assign work (plus one division and one addition): the kernels are held to the numpy statements BIT FOR
BIT, NaN payload positions and the sign of zero included."""

from __future__ import annotations

import json
import logging
import os

import numpy as np
import pytest
import torch

from anemoi_transform_amd import native
from anemoi_transform_amd.filters import create_filter_by_name, filter_registry
from oracle import oracle

import native_double
from test_filters import collect_fields_by_param, nan_rows, test_source
from test_multi_filters import mars_test_source

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))
MD = {"latitudes": [10.0, 0.0, -10.0], "longitudes": [20, 40.0], "valid_datetime": "2018-08-01T09:00:00Z"}


@pytest.fixture(params=["double", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request, monkeypatch):
    if request.param == "double":
        native_double.install(monkeypatch)
    return request.param


def same_bits(a, b):
    """Equal as bit patterns, except that any NaN equals any NaN (numpy's and the device's canonical quiet NaN agree anyway)."""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    both_nan = np.isnan(a) & np.isnan(b)
    u = {4: np.uint32, 8: np.uint64}[a.dtype.itemsize]
    return bool(np.all(both_nan | (a.view(u) == b.view(u))))


def test_the_four_filters_are_registered():
    for name in ("rodeo_opera_clipping", "rodeo_opera_preprocessing", "oras6_clipping", "land_parameters"):
        assert name in filter_registry.registered


# ---- the oracle against the reference's literals -----------------------------------------------------------------------------
def test_oracle_opera_statements_against_the_reference_vectors():
    c = GOLDEN["rodeo_opera_clipping"]
    tp, qi = oracle.opera_clipping(nan_rows(c["tp"]), nan_rows(c["qi"]), c["max_total_precipitation"])
    assert np.allclose(tp, nan_rows(c["expected_tp"]), equal_nan=True) and np.allclose(qi, nan_rows(c["expected_qi"]), equal_nan=True)
    p = GOLDEN["rodeo_opera_preprocessing"]
    tp, qi = oracle.opera_preprocessing(nan_rows(p["tp"]), nan_rows(p["qi"]), np.array(p["dm"]), p["max_total_precipitation"])
    assert np.allclose(tp, nan_rows(p["expected_tp"]), equal_nan=True) and np.allclose(qi, nan_rows(p["expected_qi"]), equal_nan=True)
    # boolean-mask assignments: -0.0 is not < 0 and stays, a NaN fails both tests and stays (R: rodeo_opera_preprocessing.py:35-36)
    kept = oracle.opera_clip_variable(np.array([-0.0, np.nan, 5.0, 7.0]), 5.0)
    assert np.signbit(kept[0]) and np.isnan(kept[1]) and kept[2] == 5.0 and kept[3] == 5.0


# ---- rodeo_opera_clipping (R: tests/field_filters/test_rodeo_opera_clipping.py) ----------------------------------------------------
def test_rodeo_opera_clipping(engine):
    g = GOLDEN["rodeo_opera_clipping"]
    source = test_source([{"param": "tp", "values": nan_rows(g["tp"]), **MD}, {"param": "qi", "values": nan_rows(g["qi"]), **MD}])
    out = collect_fields_by_param(source | create_filter_by_name("rodeo_opera_clipping", max_total_precipitation=g["max_total_precipitation"]))
    assert set(out) == {"tp", "qi"} and len(out["tp"]) == 1 and len(out["qi"]) == 1
    tp, qi = out["tp"][0].to_numpy(), out["qi"][0].to_numpy()
    assert np.allclose(tp, nan_rows(g["expected_tp"]), equal_nan=True)
    assert np.allclose(qi, nan_rows(g["expected_qi"]), equal_nan=True)
    assert np.isnan(tp).sum() == np.isnan(qi).sum()
    assert np.nanmax(tp) <= g["max_total_precipitation"] and np.nanmin(tp) >= 0.0 and np.nanmax(qi) <= 1 and np.nanmin(qi) >= 0.0
    want_tp, want_qi = oracle.opera_clipping(nan_rows(g["tp"]), nan_rows(g["qi"]), g["max_total_precipitation"])
    assert same_bits(tp, want_tp) and same_bits(qi, want_qi)


def test_rodeo_opera_clipping_default_maximum_and_two_steps(engine):
    rng = np.random.default_rng(3)
    specs, want = [], {}
    for date in (0, 6):
        tp = rng.normal(4000.0, 6000.0, (3, 2))
        qi = rng.normal(0.5, 0.6, (3, 2))
        tp[0, 0], qi[2, 1] = -0.0, np.nan
        md = dict(MD, step=date)
        specs += [{"param": "precip", "values": tp, **md}, {"param": "quality", "values": qi, **md}]
        want[date] = oracle.opera_clipping(tp, qi)  # MAX_TP = 10000 (R: rodeo_opera_clipping.py:22)
    flt = create_filter_by_name("rodeo_opera_clipping", total_precipitation="precip", quality="quality")
    out = list(mars_test_source(specs) | flt)
    assert [f.metadata("param") for f in out] == ["precip", "quality"] * 2
    for i, date in enumerate(want):
        assert same_bits(out[2 * i].to_numpy(), want[date][0]) and same_bits(out[2 * i + 1].to_numpy(), want[date][1])
    assert np.signbit(out[0].to_numpy()[0, 0])  # -0.0 / 1000 stays -0.0, as in the reference


# ---- rodeo_opera_preprocessing (R: tests/field_filters/test_rodeo_opera_preprocessing.py) ------------------------------------------
def _opera_source(g):
    return test_source([{"param": "tp", "values": nan_rows(g["tp"]), **MD}, {"param": "qi", "values": nan_rows(g["qi"]), **MD},
                        {"param": "dm", "values": np.array(g["dm"], dtype=float), **MD}])


def test_rodeo_opera_preprocessing(engine):
    g = GOLDEN["rodeo_opera_preprocessing"]
    out = collect_fields_by_param(_opera_source(g) | create_filter_by_name("rodeo_opera_preprocessing", max_total_precipitation=g["max_total_precipitation"]))
    assert set(out) == {"tp", "qi"} and len(out["tp"]) == 1 and len(out["qi"]) == 1
    tp, qi = out["tp"][0].to_numpy(), out["qi"][0].to_numpy()
    assert np.allclose(tp, nan_rows(g["expected_tp"]), equal_nan=True)
    assert np.allclose(qi, nan_rows(g["expected_qi"]), equal_nan=True)
    assert np.isnan(tp).sum() == np.isnan(qi).sum()
    assert np.nanmax(tp) <= g["max_total_precipitation"] and np.nanmin(tp) >= 0.0 and np.nanmax(qi) <= 1 and np.nanmin(qi) >= 0.0


def test_rodeo_opera_preprocessing_return_mask(engine):
    g = GOLDEN["rodeo_opera_preprocessing"]
    out = list(_opera_source(g) | create_filter_by_name("rodeo_opera_preprocessing", return_mask=True))
    assert [f.metadata("param") for f in out] == ["tp", "qi", "dm"]  # the mask follows the results (R: rodeo_opera_preprocessing.py:202-205)
    assert np.allclose(out[2].to_numpy(), np.array(g["dm"], dtype=float))
    want_tp, want_qi = oracle.opera_preprocessing(nan_rows(g["tp"]), nan_rows(g["qi"]), np.array(g["dm"], dtype=float))
    assert same_bits(out[0].to_numpy(), want_tp) and same_bits(out[1].to_numpy(), want_qi)


def test_rodeo_opera_preprocessing_warns_when_the_nan_counts_differ(engine, caplog):
    """R: rodeo_opera_preprocessing.py:91-93."""
    specs = [{"param": "tp", "values": np.array([[1.0, 2.0]]), "latitudes": [0.0], "longitudes": [0.0, 1.0]},
             {"param": "qi", "values": np.array([[0.5, 0.5]]), "latitudes": [0.0], "longitudes": [0.0, 1.0]},
             {"param": "dm", "values": np.array([[1.0, 0.0]]), "latitudes": [0.0], "longitudes": [0.0, 1.0]}]
    with caplog.at_level(logging.WARNING):
        out = collect_fields_by_param(test_source(specs) | create_filter_by_name("rodeo_opera_preprocessing"))
    assert np.isnan(out["tp"][0].to_numpy()[0, 0]) and out["qi"][0].to_numpy()[0, 0] == 0.5
    assert any("Mismatch between NaNs on tp 1 and qi 0" in r.message for r in caplog.records)


# ---- oras6_clipping (R: filters/fields/oras6_clipping.py; no test in the reference: the oracle's restatement is the authority) -------
def _oras6_arrays(rng, n, celsius, dtype=np.float64):
    a = {name: rng.normal(0.0, 1.0, n) for name in oracle.ORAS6_FIELDS}
    a["siconc"] = np.where(rng.random(n) < 0.4, rng.choice([0.0, 1e-5, 9e-6, -1e-3], n), rng.random(n))
    a["siconc"][:3] = [np.nan, 1e-5, np.nextafter(1e-5, 1.0)]
    for name in ("sihc", "snhc"):
        a[name] = -np.abs(rng.normal(0.0, 3e-5, n))  # around -PUNY, both sides
        a[name][3:6] = [-1e-5, np.nextafter(-1e-5, -1.0), np.nan]
    for name in ("sitemptop", "vasit"):
        a[name] = rng.normal(260.0, 8.0, n)
    a["sntemp"] = rng.normal(-8.0 if celsius else 265.0, 6.0, n)
    a["sntemp"][6] = np.nan
    a["tos"] = rng.normal(272.0, 1.5, n)
    a["tos"][7:10] = [271.15 - 1e-5, np.nan, -np.inf]
    return {k: v.astype(dtype) for k, v in a.items()}


@pytest.mark.parametrize("celsius", [False, True], ids=["kelvin", "celsius"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_oras6_clipping(engine, celsius, dtype):
    rng = np.random.default_rng(11 + celsius)
    n = 1500
    arrays = _oras6_arrays(rng, n, celsius, dtype)
    md = {"latitudes": np.linspace(-60, 60, n), "longitudes": np.linspace(0, 359, n)}
    specs = [{"param": "2t", "values": rng.normal(280, 5, n).astype(dtype), **md}]
    specs += [{"param": f"avg_{name}", "values": arrays[name].copy(), **md} for name in oracle.ORAS6_FIELDS]
    out = list(test_source(specs) | create_filter_by_name("oras6_clipping"))
    want = oracle.oras6_clipping(**arrays)
    assert [f.metadata("param") for f in out] == ["2t"] + [f"avg_{name}" for name, _ in want]  # unmatched first, then the reference's order
    for f, (name, values) in zip(out[1:], want):
        got = f.to_numpy()
        assert got.dtype == dtype and same_bits(got, values), name
    by_name = dict(want)
    assert (by_name["sntemp"][~np.isnan(by_name["sntemp"])] > 200).all()  # Kelvin either way
    assert not same_bits(by_name["siue"], arrays["siue"])  # the mask did something


def test_oras6_clipping_renamed_inputs_and_two_steps(engine):
    rng = np.random.default_rng(5)
    n = 64
    md = {"latitudes": np.linspace(-60, 60, n), "longitudes": np.linspace(0, 359, n)}
    names = {name: f"x_{name}" for name in oracle.ORAS6_FIELDS}
    specs, want = [], []
    for step, celsius in ((0, True), (24, False)):
        arrays = _oras6_arrays(rng, n, celsius)
        specs += [{"param": names[name], "values": arrays[name].copy(), "step": step, **md} for name in oracle.ORAS6_FIELDS]
        want += oracle.oras6_clipping(**arrays)
    out = list(mars_test_source(specs) | create_filter_by_name("oras6_clipping", **names))
    assert len(out) == 28
    for f, (name, values) in zip(out, want):  # each step decided on its own whether its snow temperature was in Celsius
        assert f.metadata("param") == names[name] and same_bits(f.to_numpy(), values), name


# ---- land_parameters (R: filters/fields/land_parameters.py; no test in the reference) ------------------------------------------------
def test_land_parameters(engine):
    rng = np.random.default_rng(2)
    n = 997
    md = {"latitudes": np.linspace(-60, 60, n), "longitudes": np.linspace(0, 359, n)}
    tvh, tvl, slt = rng.integers(0, 21, n).astype(float), rng.integers(0, 21, n).astype(float), rng.integers(0, 8, n).astype(float)
    tvh[:21] = np.arange(21)
    slt[:8] = np.arange(8)
    specs = [{"param": "tvh", "values": tvh, **md}, {"param": "tvl", "values": tvl, **md}, {"param": "slt", "values": slt, **md}]
    out = list(test_source(specs) | create_filter_by_name("land_parameters"))
    assert [f.metadata("param") for f in out] == ["hveg_rsmin", "hveg_cov", "hveg_z0m", "lveg_rsmin", "lveg_cov", "lveg_z0m", "theta_pwp", "theta_cap"]
    want = oracle.crosswalk(tvh, oracle.VEGETATION_TABLE) + oracle.crosswalk(tvl, oracle.VEGETATION_TABLE) + oracle.crosswalk(slt, oracle.SOIL_TABLE)
    for f, values in zip(out, want):
        got = f.to_numpy()
        assert got.dtype == np.float64 and same_bits(got, values), f.metadata("param")
    # the tables are the reference's (spot values of land_parameters.py:20-52)
    assert out[0].to_numpy()[3] == 395.0 and out[2].to_numpy()[18] == 1.50 and out[6].to_numpy()[5] == 0.335 and out[7].to_numpy()[6] == 0.663


@pytest.mark.parametrize("bad", [21.0, -1.0, 2.5, np.nan], ids=["beyond", "negative", "fraction", "nan"])
def test_land_parameters_refuses_a_class_that_is_no_key(engine, bad):
    """R: land_parameters.py:71 `param_dic[x]` raises KeyError."""
    n = 16
    md = {"latitudes": np.linspace(-60, 60, n), "longitudes": np.linspace(0, 359, n)}
    tvh = np.full(n, 3.0)
    tvh[5] = bad
    specs = [{"param": "tvh", "values": tvh, **md}, {"param": "tvl", "values": np.zeros(n), **md}, {"param": "slt", "values": np.ones(n), **md}]
    with pytest.raises(KeyError):
        list(test_source(specs) | create_filter_by_name("land_parameters"))
    with pytest.raises(KeyError):
        oracle.crosswalk(tvh, oracle.VEGETATION_TABLE)


def test_land_parameters_float32_classes_and_renamed_outputs(engine):
    n = 40
    md = {"latitudes": np.linspace(-60, 60, n), "longitudes": np.linspace(0, 359, n)}
    cls = (np.arange(n) % 8).astype(np.float32)
    specs = [{"param": "hv", "values": cls, **md}, {"param": "lv", "values": cls, **md}, {"param": "soil", "values": cls, **md}]
    flt = create_filter_by_name("land_parameters", high_veg_type="hv", low_veg_type="lv", soil_type="soil", theta_cap="field_capacity")
    out = collect_fields_by_param(test_source(specs) | flt)
    got = out["field_capacity"][0].to_numpy()
    assert got.dtype == np.float64 and same_bits(got, oracle.crosswalk(cls, oracle.SOIL_TABLE)[1])  # np.array of Python floats: float64


# ---- the kernels, directly, against the oracle on large seeded inputs (GPU) ------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("np_dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("layout", [native.COLUMNS, native.FIELDS], ids=["columns", "fields"])
def test_domain_kernels_bit_for_bit(dev, np_dtype, layout):
    from anemoi_transform_amd.stack import Stack

    rng = np.random.default_rng(31)
    n_lev, n_pts = 5, 20011
    tdtype = torch.float64 if np_dtype == np.float64 else torch.float32

    def special(x):
        x = x.astype(np_dtype)
        x[:, :8] = np.array([-0.0, 0.0, np.nan, np.inf, -np.inf, 1.0, 12.5, np.nextafter(np_dtype(12.5), np_dtype(0))], dtype=np_dtype)
        return x

    def run(op, ins, n_out, level_param, shared=None):
        stacks = [Stack.from_fields(x, dev=dev, layout=layout) for x in ins]
        outs = [stacks[0].new_like(zero=False) for _ in range(n_out)]
        tensors = [s.data for s in stacks] + ([] if shared is None else [torch.from_numpy(shared).to(dev)])
        native.combine_stack(op, tensors, [o.data for o in outs], n_pts=n_pts, n_lev=n_lev, pitch=stacks[0].pitch, layout=layout,
                             level_param=torch.tensor(level_param, dtype=torch.float64, device=dev))
        return [o.numpy() for o in outs]

    tp, qi = special(rng.normal(5.0, 10.0, (n_lev, n_pts))), special(rng.normal(0.5, 0.6, (n_lev, n_pts)))
    dm = rng.integers(0, 5, (n_lev, n_pts)).astype(np_dtype)
    limits = [12.5, 10000.0, 1.0, 0.0, 3.0]
    got = run(native.COMB_OPERA_CLIP, [tp, qi], 2, limits)
    for l in range(n_lev):
        want = oracle.opera_clipping(tp[l], qi[l], limits[l])
        assert same_bits(got[0][l], want[0]) and same_bits(got[1][l], want[1]), l
    got = run(native.COMB_OPERA_PREPROCESS, [tp, qi, dm], 2, limits)
    for l in range(n_lev):
        want = oracle.opera_preprocessing(tp[l], qi[l], dm[l], limits[l])
        assert same_bits(got[0][l], want[0]) and same_bits(got[1][l], want[1]), l

    # oras6: every kind of level against the statement of the field it stands for
    siconc = np.where(rng.random(n_pts) < 0.5, rng.choice([0.0, 1e-5, 9e-6, -1.0], n_pts), rng.random(n_pts)).astype(np_dtype)
    siconc[:4] = [np.nan, 1e-5, np.nextafter(np_dtype(1e-5), np_dtype(1)), 0.0]
    kinds = [native.ORAS6_ZERO, native.ORAS6_TEMPERATURE, native.ORAS6_CELSIUS, native.ORAS6_HEAT, native.ORAS6_SURFACE]
    x = special(rng.normal(0.0, 1.0, (n_lev, n_pts)))
    x[3] = special(-np.abs(rng.normal(0.0, 3e-5, (1, n_pts))))[0]
    x[4] = special(rng.normal(271.5, 1.0, (1, n_pts)))[0]
    (got,) = run(native.COMB_ORAS6, [x], 1, [float(k) for k in kinds], shared=siconc)
    for l, kind in enumerate(kinds):
        assert same_bits(got[l], native_double._oras6_level(x[l].copy(), siconc.copy(), kind)), kind
    (kept,) = run(native.COMB_ORAS6, [x], 1, [float(native.ORAS6_KEEP)] * n_lev, shared=siconc)
    assert same_bits(kept, x)

    # look-up
    classes = rng.integers(0, 21, (n_lev, n_pts)).astype(np_dtype)
    classes[0, :5] = [21.0, -1.0, 2.5, np.nan, 20.0]
    table = [oracle.VEGETATION_TABLE[c][2] for c in range(21)]
    (got,) = run(native.COMB_LOOKUP, [classes], 1, [21.0] + table)
    assert np.isnan(got[0, :4]).all() and got[0, 4] == np_dtype(0.02)
    ok = np.ones_like(classes, dtype=bool)
    ok[0, :4] = False
    assert np.array_equal(got[ok], np.array(table)[classes[ok].astype(int)].astype(np_dtype))
    assert got.dtype == np_dtype and tdtype in (torch.float32, torch.float64)
