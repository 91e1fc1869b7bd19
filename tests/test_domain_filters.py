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


# =================================================================================
# humidity conversions (R: tests/field_filters/test_dewpoint.py, test_pressure_level_humidity.py, test_q_height_with_p.py)
# =================================================================================
def arr(x):
    return np.array(x, dtype=np.float64)


def fields_equal(a, b):
    """R: tests/utils/__init__.py:26-55 `assert_fields_equal`, the keys these tests use."""
    for key in ("param", "valid_datetime", "levelist"):
        assert a.metadata(key, default=None) == b.metadata(key, default=None)
    assert np.allclose(a.to_numpy(), b.to_numpy(), equal_nan=True)
    return True


def select(fields, params):
    """R: tests/utils/__init__.py `SelectFieldSource`: the fields of some params, as a source."""
    return test_source_from([f for f in fields if f.metadata("param") in params])


def test_source_from(fields):
    from anemoi_transform_amd.core import source_registry
    from anemoi_transform_amd.fields import FieldList

    return source_registry.create("testing", dataset=FieldList(list(fields)))


test_source_from.__test__ = False


def test_oracle_humidity_statements_against_the_reference_vectors():
    """The restated earthkit-meteo functions against the reference's literals, at its tests' own tolerance (np.allclose defaults)."""
    d = GOLDEN["dewpoint"]
    assert np.allclose(oracle.dewpoint_from_relative_humidity(arr(d["r"]), arr(d["t"])), arr(d["d"]))
    assert np.allclose(oracle.relative_humidity_from_dewpoint(arr(d["d"]), arr(d["t"])), arr(d["r"]))
    h = GOLDEN["pressure_level_humidity"]
    for level in h["levels"]:
        t, q, r = (arr(h[k][str(level)]) for k in ("t", "q", "r"))
        assert np.allclose(oracle.relative_humidity_from_specific_humidity(t, q, 100.0 * level), r)
        assert np.allclose(oracle.specific_humidity_from_relative_humidity(t, r, 100.0 * level), q)
    # both phases are exercised by those vectors: 248.9 K and 250.15 K lie below Ti = 250.16 K (ice only), 260.5 K and 271.3 K in between
    assert (arr(h["t"]["850"]) < oracle.MET_TI).any() and ((arr(h["t"]["850"]) > oracle.MET_TI) & (arr(h["t"]["850"]) < oracle.MET_T0)).any()
    # the guard of specific_humidity_from_vapour_pressure and the zero guard of the dewpoint filter
    assert np.isnan(oracle.specific_humidity_from_relative_humidity(np.array([373.0]), np.array([200.0]), 50000.0)[0])
    assert np.isfinite(oracle.dewpoint_from_relative_humidity(np.array([0.0]), np.array([280.0]))[0])


def test_relative_humidity_to_dewpoint(engine):
    """R: test_dewpoint.py:49-70."""
    g = GOLDEN["dewpoint"]
    src = test_source([{"param": "r", "values": arr(g["r"]), **MD}, {"param": "t", "values": arr(g["t"]), **MD}])
    inputs, out = collect_fields_by_param(src), collect_fields_by_param(src | create_filter_by_name("r_to_d"))
    assert set(out) == {"r", "t", "d"} and len(out["d"]) == 1
    assert all(fields_equal(inputs[p][0], out[p][0]) for p in ("r", "t"))
    assert np.allclose(out["d"][0].to_numpy(), arr(g["d"]))
    np.testing.assert_allclose(out["d"][0].to_numpy(), oracle.dewpoint_from_relative_humidity(arr(g["r"]), arr(g["t"])), rtol=1e-13)


def test_dewpoint_to_relative_humidity_and_round_trip(engine):
    """R: test_dewpoint.py:73-97, :127-146."""
    g = GOLDEN["dewpoint"]
    dsrc = test_source([{"param": "d", "values": arr(g["d"]), **MD}, {"param": "t", "values": arr(g["t"]), **MD}])
    out = collect_fields_by_param(dsrc | create_filter_by_name("d_to_r"))
    assert set(out) == {"d", "t", "r"} and np.allclose(out["r"][0].to_numpy(), arr(g["r"]))
    src = test_source([{"param": "r", "values": arr(g["r"]), **MD}, {"param": "t", "values": arr(g["t"]), **MD}])
    mid = list(src | create_filter_by_name("r_to_d"))
    back = collect_fields_by_param(select(mid, ["t", "d"]) | create_filter_by_name("d_to_r"))  # r dropped: it has to be reconstructed
    assert set(back) == {"r", "d", "t"}
    assert np.allclose(back["r"][0].to_numpy(), arr(g["r"])) and fields_equal(back["t"][0], collect_fields_by_param(src)["t"][0])


def test_zero_relative_humidity_is_guarded(engine):
    """R: dewpoint.py:62 — r == 0 becomes 1e-4 before the dewpoint is taken (finite result)."""
    r, t = np.array([[0.0, 50.0]]), np.array([[280.0, 280.0]])
    md = {"latitudes": [0.0], "longitudes": [0.0, 1.0]}
    out = collect_fields_by_param(test_source([{"param": "r", "values": r, **md}, {"param": "t", "values": t, **md}]) | create_filter_by_name("r_to_d", return_inputs="none"))
    assert set(out) == {"d"}
    got = out["d"][0].to_numpy()
    assert np.isfinite(got).all() and got[0, 0] < got[0, 1] < 280.0
    np.testing.assert_allclose(got, oracle.dewpoint_from_relative_humidity(r, t), rtol=1e-13)


def _humidity_specs(g, name):
    return [{"param": p, "levelist": level, "values": arr(g[key][str(level)]), **MD} for level in g["levels"] for p, key in ((name, name), ("t", "t"))]


def test_pressure_level_specific_humidity_to_relative_humidity(engine):
    """R: test_pressure_level_humidity.py:65-86."""
    g = GOLDEN["pressure_level_humidity"]
    src = test_source(_humidity_specs(g, "q"))
    inputs, out = collect_fields_by_param(src), collect_fields_by_param(src | create_filter_by_name("q_to_r"))
    assert set(out) == {"q", "t", "r"}
    for p in ("q", "t"):
        assert all(fields_equal(a, b) for a, b in zip(inputs[p], out[p]))
    by_level = {f.metadata("levelist"): f.to_numpy() for f in out["r"]}
    assert set(by_level) == {850, 1000}
    for level, got in by_level.items():
        assert np.allclose(got, arr(g["r"][str(level)]))
        want = oracle.relative_humidity_from_specific_humidity(arr(g["t"][str(level)]), arr(g["q"][str(level)]), 100.0 * level)
        np.testing.assert_allclose(got, want, rtol=1e-13)


def test_pressure_level_relative_humidity_to_specific_humidity_and_round_trips(engine):
    """R: test_pressure_level_humidity.py:89-114, :141-190."""
    g = GOLDEN["pressure_level_humidity"]
    rsrc = test_source(_humidity_specs(g, "r"))
    out = collect_fields_by_param(rsrc | create_filter_by_name("r_to_q"))
    assert set(out) == {"r", "t", "q"}
    for f in out["q"]:
        assert np.allclose(f.to_numpy(), arr(g["q"][str(f.metadata("levelist"))]))
    qsrc = test_source(_humidity_specs(g, "q"))
    mid = list(qsrc | create_filter_by_name("q_to_r"))
    back = collect_fields_by_param(select(mid, ["r", "t"]) | create_filter_by_name("r_to_q"))
    assert set(back) == {"q", "t", "r"}
    for f in back["q"]:
        assert np.allclose(f.to_numpy(), arr(g["q"][str(f.metadata("levelist"))]))
    with pytest.raises(KeyError):  # R: q_to_r.py:72 `humidity.metadata("levelist")` of a field without a level
        list(test_source([{"param": "q", "values": arr(g["q"]["850"]), **MD}, {"param": "t", "values": arr(g["t"]["850"]), **MD}]) | create_filter_by_name("q_to_r"))


def test_q_to_r_height_with_p_and_round_trip(engine):
    """R: test_q_height_with_p.py:57-110 (its expected values are computed by earthkit-meteo there; here by the oracle's restatement)."""
    g = GOLDEN["height_level_humidity_with_p"]
    t, q, p = arr(g["t"]), arr(g["q"]), arr(g["p"])
    src = test_source([{"param": "q", "values": q.copy(), **MD}, {"param": "t", "values": t.copy(), **MD}, {"param": "pres", "values": p.copy(), **MD}])
    inputs, out = collect_fields_by_param(src), collect_fields_by_param(src | create_filter_by_name("q_to_r_height_with_p"))
    assert set(out) == {"q", "t", "pres", "r"}
    assert all(fields_equal(inputs[k][0], out[k][0]) for k in ("q", "t", "pres"))
    want = oracle.relative_humidity_from_specific_humidity(t, q, p)
    np.testing.assert_allclose(out["r"][0].to_numpy(), want, rtol=1e-13)
    assert (want > 20).all() and (want < 140).all()
    mid = list(src | create_filter_by_name("q_to_r_height_with_p"))
    back = collect_fields_by_param(select(mid, ["r", "t", "pres"]) | create_filter_by_name("r_to_q_height_with_p"))
    assert set(back) == {"q", "t", "pres", "r"}
    np.testing.assert_allclose(back["q"][0].to_numpy(), q)  # the reference's round-trip tolerance (assert_allclose defaults)


@pytest.mark.gpu
@pytest.mark.parametrize("np_dtype,rtol", [(np.float64, 2e-13), (np.float32, 2e-5)], ids=["f64", "f32"])
@pytest.mark.parametrize("layout", [native.COLUMNS, native.FIELDS], ids=["columns", "fields"])
def test_humidity_kernels_vs_oracle(dev, np_dtype, rtol, layout):
    """Floating point with exp / log from the device library: held to the numpy restatement within a few ulp of the result
    (float64 2e-13, float32 2e-5 relative) on the ranges the atmosphere has, all three phases of the saturation curve included."""
    from anemoi_transform_amd.stack import Stack

    rng = np.random.default_rng(41)
    n_lev, n_pts = 6, 30011
    t = rng.uniform(190.0, 320.0, (n_lev, n_pts)).astype(np_dtype)
    t[:, :4] = np.array([250.16, 273.16, 250.15999, 273.16001], dtype=np_dtype)  # the ends of the mixed phase
    r = rng.uniform(0.5, 110.0, (n_lev, n_pts)).astype(np_dtype)
    r[0, 4:6] = [0.0, 100.0]
    q = (10.0 ** rng.uniform(-6, -1.7, (n_lev, n_pts))).astype(np_dtype)
    p = rng.uniform(20000.0, 105000.0, (n_lev, n_pts)).astype(np_dtype)
    levels = np.array([50.0, 200.0, 500.0, 700.0, 850.0, 1000.0])

    def run(op, ins, level_param=None):
        stacks = [Stack.from_fields(x, dev=dev, layout=layout) for x in ins]
        out = stacks[0].new_like(zero=False)
        lp = None if level_param is None else torch.from_numpy(level_param).to(dev)
        native.combine_stack(op, [s.data for s in stacks], [out.data], n_pts=n_pts, n_lev=n_lev, pitch=stacks[0].pitch, layout=layout, level_param=lp)
        return out.numpy()

    def close(got, want):
        assert got.dtype == np_dtype
        ok = np.isfinite(want)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        np.testing.assert_allclose(got[ok], want[ok].astype(np_dtype), rtol=rtol)

    td = oracle.dewpoint_from_relative_humidity(r, t).astype(np_dtype)
    close(run(native.COMB_R_TO_D, [r, t]), td)
    close(run(native.COMB_D_TO_R, [td, t]), oracle.relative_humidity_from_dewpoint(td, t))
    close(run(native.COMB_Q_TO_R, [q, t, p]), oracle.relative_humidity_from_specific_humidity(t, q, p))
    close(run(native.COMB_R_TO_Q, [r, t, p]), oracle.specific_humidity_from_relative_humidity(t, r, p))
    want = np.stack([oracle.relative_humidity_from_specific_humidity(t[l], q[l], np_dtype(100.0) * np_dtype(levels[l])) for l in range(n_lev)])
    close(run(native.COMB_Q_TO_R, [q, t], levels), want)
    want = np.stack([oracle.specific_humidity_from_relative_humidity(t[l], r[l], np_dtype(100.0) * np_dtype(levels[l])) for l in range(n_lev)])
    got = run(native.COMB_R_TO_Q, [r, t], levels)
    close(got, want)
    assert np.isnan(want).any()  # hot and humid at 50 hPa: e reaches p, the guard gives NaN on both sides
    with pytest.raises(ValueError):
        run(native.COMB_Q_TO_R, [q, t])  # two operands and no level_param


# ---- random shapes: odd sizes, thin stacks (scalar path), tails, both layouts ------------------------------------------------------
# ATX_DOMAIN_SEEDS=first:count widens the sweep for a one-off soak run
_FIRST, _COUNT = (int(v) for v in os.environ.get("ATX_DOMAIN_SEEDS", "0:0").split(":"))
DOMAIN_SEEDS = range(_FIRST, _FIRST + _COUNT) if _COUNT else range(12)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", DOMAIN_SEEDS)
def test_domain_and_humidity_kernels_on_random_shapes(dev, seed):
    from anemoi_transform_amd.stack import Stack

    rng = np.random.default_rng(1000 + seed)
    np_dtype = [np.float64, np.float32][int(rng.integers(2))]
    layout = [native.COLUMNS, native.FIELDS][int(rng.integers(2))]
    n_lev = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 14, 33, 137]))
    n_pts = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 255, 257, 1000, 4097, 20001]))
    rtol = 2e-13 if np_dtype == np.float64 else 2e-5

    def stacks_of(arrays):
        return [Stack.from_fields(a, dev=dev, layout=layout) for a in arrays]

    def run(op, ins, n_out, level_param=None, shared=None):
        st = stacks_of(ins)
        outs = [st[0].new_like(zero=False) for _ in range(n_out)]
        for o in outs:
            o.data.fill_(float("nan"))  # whatever is not written shows
        tensors = [s.data for s in st] + ([] if shared is None else [torch.from_numpy(shared).to(dev)])
        lp = None if level_param is None else torch.tensor(np.asarray(level_param, dtype=np.float64), device=dev)
        native.combine_stack(op, tensors, [o.data for o in outs], n_pts=n_pts, n_lev=n_lev, pitch=st[0].pitch, layout=layout, level_param=lp)
        for o in outs:  # the padding of the outputs is written with zeros (atx.h)
            full = o.data.cpu().numpy()
            pad = full[:, n_lev:] if layout == native.COLUMNS else full[:, n_pts:]
            assert not np.isnan(pad).any() and not pad.any()
        return [o.numpy() for o in outs]

    def values(loc, scale):
        x = rng.normal(loc, scale, (n_lev, n_pts)).astype(np_dtype)
        flat = x.reshape(-1)
        k = min(flat.size, 5)
        flat[:k] = np.array([-0.0, np.nan, np.inf, 0.0, -np.inf], dtype=np_dtype)[:k]
        return x

    limits = rng.choice([0.0, 1.0, 12.5, 10000.0], n_lev)
    tp, qi = values(5.0, 10.0), values(0.5, 0.6)
    dm = rng.integers(0, 5, (n_lev, n_pts)).astype(np_dtype)
    got = run(native.COMB_OPERA_CLIP, [tp, qi], 2, limits)
    want = [oracle.opera_clipping(tp[l], qi[l], limits[l]) for l in range(n_lev)]
    assert same_bits(got[0], np.stack([w[0] for w in want])) and same_bits(got[1], np.stack([w[1] for w in want]))
    got = run(native.COMB_OPERA_PREPROCESS, [tp, qi, dm], 2, limits)
    want = [oracle.opera_preprocessing(tp[l], qi[l], dm[l], limits[l]) for l in range(n_lev)]
    assert same_bits(got[0], np.stack([w[0] for w in want])) and same_bits(got[1], np.stack([w[1] for w in want]))

    siconc = np.where(rng.random(n_pts) < 0.5, rng.choice([0.0, 1e-5, 9e-6, -1.0, np.nan], n_pts), rng.random(n_pts)).astype(np_dtype)
    kinds = rng.integers(0, 6, n_lev)
    x = values(0.0, 3e-5)
    (got,) = run(native.COMB_ORAS6, [x], 1, kinds.astype(float), shared=siconc)
    assert same_bits(got, np.stack([native_double._oras6_level(x[l].copy(), siconc.copy(), int(kinds[l])) for l in range(n_lev)]))

    classes = rng.integers(-1, 9, (n_lev, n_pts)).astype(np_dtype)
    table = rng.normal(0, 1, 8)
    (got,) = run(native.COMB_LOOKUP, [classes], 1, [8.0] + list(table))
    known = (classes >= 0) & (classes < 8)
    assert np.isnan(got[~known]).all() and np.array_equal(got[known], table[classes[known].astype(int)].astype(np_dtype))

    t = rng.uniform(190.0, 320.0, (n_lev, n_pts)).astype(np_dtype)
    r = rng.uniform(0.0, 110.0, (n_lev, n_pts)).astype(np_dtype)
    q = (10.0 ** rng.uniform(-6, -1.7, (n_lev, n_pts))).astype(np_dtype)
    p = rng.uniform(20000.0, 105000.0, (n_lev, n_pts)).astype(np_dtype)
    levels = rng.choice([50.0, 200.0, 500.0, 850.0, 1000.0], n_lev)

    def close(got, want):
        want = np.asarray(want).astype(np_dtype)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = np.isfinite(want)
        np.testing.assert_allclose(got[ok], want[ok], rtol=rtol)

    close(run(native.COMB_R_TO_D, [r, t], 1)[0], oracle.dewpoint_from_relative_humidity(r, t))
    close(run(native.COMB_D_TO_R, [t - np_dtype(3.0), t], 1)[0], oracle.relative_humidity_from_dewpoint(t - np_dtype(3.0), t))
    close(run(native.COMB_Q_TO_R, [q, t, p], 1)[0], oracle.relative_humidity_from_specific_humidity(t, q, p))
    close(run(native.COMB_R_TO_Q, [r, t], 1, levels)[0],
          np.stack([oracle.specific_humidity_from_relative_humidity(t[l], r[l], np_dtype(100.0) * np_dtype(levels[l])) for l in range(n_lev)]))
