"""Pins the CPU oracle to the reference's own test vectors (tests/golden/reference_vectors.json).

Expectations are rebuilt here from the rules the reference tests state, with plain
numpy indexing that does not go through the oracle, then compared with the oracle's
filter-level restatements.
"""

from __future__ import annotations

import itertools
import json
import os

import numpy as np
import pytest

from oracle import oracle

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))


def arr(x):
    return np.array([[np.nan if v is None else v for v in row] for row in x], dtype=np.float64) if isinstance(x[0], list) else np.array(
        [np.nan if v is None else v for v in x], dtype=np.float64
    )


def dict_fields(values: dict, metadata: dict):
    lat, lon = np.meshgrid(np.array(metadata["latitudes"], float), np.array(metadata["longitudes"], float), indexing="ij")
    return [dict(param=p, values=arr(v), latitudes=lat.ravel(), longitudes=lon.ravel()) for p, v in values.items()]


def by_param(fields):
    out = {}
    for f in fields:
        out.setdefault(f["param"], []).append(f)
    return out


G = GOLDEN["apply_mask"]


@pytest.mark.parametrize("mask_name", list(G["mask_values"]))
@pytest.mark.parametrize("opts", G["threshold_options"], ids=str)
@pytest.mark.parametrize("rename", G["rename"])
def test_apply_mask_vectors(mask_name, opts, rename):
    """R: tests/field_filters/test_apply_mask.py:63-106."""
    fields = dict_fields(G["data_values"], G["metadata"])
    mask_values = np.array(G["mask_values"][mask_name]).flatten()
    out = by_param(oracle.filter_apply_mask(fields, mask_values=mask_values, rename=rename, **opts))
    if "mask_value" in opts:
        expected_mask = mask_values == opts["mask_value"]
    else:
        expected_mask = {"<": np.less, ">": np.greater}[opts["threshold_operator"]](mask_values, opts["threshold"])
    for param, values in G["data_values"].items():
        result_param = f"{param}_{rename}" if rename else param
        assert result_param in out
        expected = np.array(values, float).flatten()
        expected[expected_mask] = np.nan
        result = out[result_param][0]["values"]
        assert np.array_equal(expected, result, equal_nan=True)
        assert np.sum(np.isnan(result)) == np.sum(expected_mask)


def test_apply_mask_single_param():
    """R: tests/field_filters/test_apply_mask.py:109-139."""
    case = G["single_param_case"]
    fields = dict_fields(G["data_values"], G["metadata"])
    mask_values = np.array(G["mask_values"][case["path"]]).flatten()
    out = by_param(oracle.filter_apply_mask(fields, mask_values=mask_values, threshold=case["threshold"],
                                            threshold_operator=case["threshold_operator"], param=case["param"]))
    expected_mask = mask_values > 0.5
    for param, values in G["data_values"].items():
        flat = np.array(values, float).flatten()
        if param == "t":
            flat[expected_mask] = np.nan
            assert np.array_equal(out[param][0]["values"].ravel(), flat, equal_nan=True)
        else:
            assert np.array_equal(np.asarray(out[param][0]["values"]).ravel(), flat)


F = GOLDEN["apply_mask_from_field"]


@pytest.mark.parametrize("case", F["cases"], ids=lambda c: str(c["options"]))
def test_apply_mask_from_field_vectors(case):
    """R: tests/field_filters/test_apply_mask_from_field.py:40-134."""
    fields = dict_fields(F["data_values"], F["metadata"])
    out = by_param(oracle.filter_apply_mask(fields, **case["options"]))
    lsm = np.array(F["lsm"]).flatten()
    expected_mask = (lsm == 0) if case["mask_rule"] == "lsm == 0" else (lsm < 0.5)
    assert ("lsm" in out) == case["lsm_in_output"]
    for param in case["masked"]:
        expected = np.array(F["data_values"][param], float).flatten()
        expected[expected_mask] = np.nan
        assert np.array_equal(out[param][0]["values"], expected, equal_nan=True)
    for param in case.get("unchanged", []):
        assert np.array_equal(np.asarray(out[param][0]["values"]).ravel(), np.array(F["data_values"][param], float).flatten())
    if case["lsm_in_output"]:
        assert np.array_equal(np.asarray(out["lsm"][0]["values"]).ravel(), lsm)


@pytest.mark.parametrize("err", F["errors"], ids=lambda e: e["match"])
def test_apply_mask_errors(err):
    fields = dict_fields(F["data_values"], F["metadata"])
    opts = dict(err["options"])
    if "path" in opts:
        opts["mask_values"] = np.zeros(6)
        del opts["path"]
    with pytest.raises(ValueError, match=err["match"]):
        oracle.filter_apply_mask(fields, **opts)


def test_remove_nans_vectors():
    """R: tests/field_filters/test_remove_nans.py:23-45,77-98 — literal expected values and lat/lon."""
    R = GOLDEN["remove_nans"]
    lat, lon = np.meshgrid(np.array(R["metadata"]["latitudes"]), np.array(R["metadata"]["longitudes"]), indexing="ij")
    fields = [dict(param="t", step=i, values=arr(v), latitudes=lat.ravel(), longitudes=lon.ravel()) for i, v in enumerate(R["input_values"])]
    out = oracle.filter_remove_nans(fields)
    for f, expected in zip(out, R["expected_values"]):
        assert np.array_equal(f["values"], arr(expected), equal_nan=True)
        assert np.array_equal(f["latitudes"], R["expected_latitudes"])
        assert np.array_equal(f["longitudes"], R["expected_longitudes"])


def test_remove_nans_param_choice():
    """R: tests/field_filters/test_remove_nans.py:48-72,113-130."""
    R = GOLDEN["remove_nans"]
    lat, lon = np.meshgrid(np.array(R["metadata"]["latitudes"]), np.array(R["metadata"]["longitudes"]), indexing="ij")
    mk = lambda p, vs: [dict(param=p, step=i, values=arr(v), latitudes=lat.ravel(), longitudes=lon.ravel()) for i, v in enumerate(vs)]
    fields = mk("t", R["input_values"]) + mk("a", R["input_values"][::-1])
    out = {p: by_param(oracle.filter_remove_nans(fields, param=p)) for p in ("a", "t", None)}
    assert out["a"]["a"][0]["values"].shape != out["t"]["a"][0]["values"].shape
    assert out["t"]["t"][0]["values"].shape == out[None]["t"][0]["values"].shape
    for i in range(3):
        assert out[None]["a"][i]["values"].shape == out[None]["t"][i]["values"].shape


def test_orog_to_z_vectors():
    """R: tests/field_filters/test_orog_to_z.py:25-27,42-97."""
    O = GOLDEN["orog_to_z"]
    assert oracle.G == GOLDEN["constants"]["g"]
    orog = np.array(O["orog"])
    fields = dict_fields({"orog": O["orog"]}, O["metadata"])
    z = oracle.filter_orog_to_z(fields)
    assert [f["param"] for f in z] == ["z"]
    assert np.allclose(z[0]["values"], orog * GOLDEN["constants"]["g"])
    back = oracle.filter_orog_to_z(z, backward=True)
    assert [f["param"] for f in back] == ["orog"]
    assert np.allclose(back[0]["values"], orog)


def test_lnsp_vectors():
    """R: tests/field_filters/test_lnsp_to_sp.py:24-48."""
    L = GOLDEN["lnsp_to_sp"]
    fields = dict_fields({"lnsp": L["lnsp"]}, L["metadata"])
    sp = oracle.filter_lnsp_to_sp(fields)
    assert sp[0]["param"] == "sp" and np.allclose(sp[0]["values"], np.exp(np.array(L["lnsp"])))
    back = oracle.filter_lnsp_to_sp(sp, backward=True)
    assert back[0]["param"] == "lnsp" and np.allclose(back[0]["values"], np.array(L["lnsp"]))


@pytest.mark.parametrize("params,value", [("t", 0.0), (["t", "q"], -1.0), ("r", 0.0)])
def test_impute_nans_vectors(params, value):
    """R: tests/field_filters/test_impute_nans.py:38-88."""
    I = GOLDEN["impute_nans"]
    fields = dict_fields({k: I[k] for k in ("t", "q", "r")}, I["metadata"])
    out = by_param(oracle.filter_impute_nans(fields, param=params, value=value))
    selected = [params] if isinstance(params, str) else params
    for p in ("t", "q", "r"):
        original = arr(I[p]).flatten()
        result = np.asarray(out[p][0]["values"]).ravel()
        if p in selected:
            expected = original.copy()
            expected[np.isnan(expected)] = value
            assert np.array_equal(result, expected) and not np.any(np.isnan(result))
        else:
            assert np.array_equal(result, original, equal_nan=True)


def test_rescale_k_to_degc():
    """R: tests/field_filters/test_rescale.py:58-72: K -> degC is x - 273.15, and back."""
    c = GOLDEN["rescale"]["K_to_degC"]
    x = np.array([250.0, 273.15, 300.0])
    f = [dict(param="2t", values=x, latitudes=np.zeros(3), longitudes=np.zeros(3))]
    y = oracle.filter_rescale(f, scale=c["scale"], offset=c["offset"], param="2t")
    np.testing.assert_allclose(y[0]["values"], x - 273.15)
    np.testing.assert_allclose(oracle.filter_rescale(y, scale=c["scale"], offset=c["offset"], param="2t", backward=True)[0]["values"], x)


def test_interpolator_dispatch_order():
    """R: filters/fields/regrid.py:432-467."""
    assert oracle.interpolator_name(matrix="m.npz", mask="k.npz", method="nearest") == "MIRMatrix"
    assert oracle.interpolator_name(mask="k.npz", method="nearest") == "MaskedRegrid"
    assert oracle.interpolator_name(method="nearest") == "ScipyKDTreeNearestNeighbours"
    assert oracle.interpolator_name(method="linear") == "EarthkitRegrid"
    assert oracle.interpolator_name() == "EarthkitRegrid"


def test_nearest_grid_points_contract():
    """R: spatial.py:587-635: exact hits, wrap-around in longitude, return order (indices, distances)."""
    src_lat = np.array([0.0, 0.0, 45.0, -45.0, 89.0])
    src_lon = np.array([0.0, 180.0, 90.0, 270.0, 10.0])
    idx = oracle.nearest_grid_points(src_lat, src_lon, src_lat, src_lon)
    assert idx.dtype == np.int64 and np.array_equal(idx, np.arange(5))
    idx, dist = oracle.nearest_grid_points(src_lat, src_lon, np.array([1.0, 0.5]), np.array([359.0, 181.0]),
                                           num_neighbours_to_return=2, return_distances=True)
    assert idx.shape == (2, 2) and np.array_equal(idx[:, 0], [0, 1])
    assert np.all(np.diff(dist, axis=1) >= 0)
    # chord distance on the unit sphere
    assert dist[0, 0] == pytest.approx(2 * np.sin(np.deg2rad(np.sqrt(2.0)) / 2), rel=1e-3)
    # "no neighbour": index == len(source) (R: spatial.py:630-632)
    far = oracle.nearest_grid_points(src_lat, src_lon, np.array([0.0]), np.array([90.0]), max_distance=1e-3)
    assert far[0] == len(src_lat)


def test_csr_and_gather_statements():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(50)
    idx = rng.integers(0, 50, size=20)
    assert np.array_equal(oracle.gather_nn(x, idx), x[idx])
    w = rng.random((20, 3))
    cols = rng.integers(0, 50, size=(20, 3))
    want = (w * x[cols]).sum(axis=1)
    got = oracle.csr_apply(w.ravel(), cols.ravel(), np.arange(21) * 3, (20, 50), x)
    np.testing.assert_allclose(got, want, rtol=1e-14)
    m = rng.random(50) < 0.5
    assert np.array_equal(oracle.masked_subset(x, m), x[m])
    assert np.array_equal(oracle.masked_subset(x, np.flatnonzero(m)), x[m])


# ---- multi-input statements (R: tests/field_filters/test_cos_sin_*.py, test_sum.py, test_snow_depth_m.py, test_glacier_mask.py,
#      test_accum_to_interval.py) -------------------------------------------------------------------------------------------
def test_cos_sin_vectors():
    g = GOLDEN["cos_sin_from_rad"]
    cos, sin = oracle.cos_sin(arr(g["rad"]))
    np.testing.assert_allclose(cos, arr(g["cos"]))  # the reference's tolerance: assert_allclose defaults (rtol 1e-7)
    np.testing.assert_allclose(sin, arr(g["sin"]))
    np.testing.assert_allclose(oracle.direction_from_cos_sin(arr(g["cos"]), arr(g["sin"])), arr(g["rad"]))
    d = GOLDEN["cos_sin_mean_wave_direction"]
    cos, sin = oracle.cos_sin(arr(d["mwd"]), degrees=True)
    np.testing.assert_allclose(cos, arr(d["cos"]))
    np.testing.assert_allclose(sin, arr(d["sin"]))
    np.testing.assert_allclose(oracle.direction_from_cos_sin(arr(d["cos"]), arr(d["sin"]), degrees=True), arr(d["mwd"]))


def test_uv_to_ddff_vectors():
    """The restated earthkit-meteo wind conversions against the reference's own expectations (np.allclose, as its tests use)."""
    g = GOLDEN["uv_to_ddff"]
    for level in g["levels"]:
        u, v, ws, wdir = (arr(g[k][str(level)]) for k in ("u", "v", "ws", "wdir"))
        speed, direction = oracle.xy_to_polar(u, v)
        assert np.allclose(speed, ws) and np.allclose(direction, wdir)
        assert direction.min() >= 0.0 and direction.max() < 360.0
        back_u, back_v = oracle.polar_to_xy(ws, wdir)
        assert np.allclose(back_u, u) and np.allclose(back_v, v)
        rt_u, rt_v = oracle.polar_to_xy(*oracle.xy_to_polar(u, v))
        assert np.allclose(rt_u, u) and np.allclose(rt_v, v)


def test_sum_vectors():
    g = GOLDEN["sum"]
    t, r = arr(g["t"]), arr(g["r"])
    assert np.allclose(oracle.sum_fields([r.copy(), t.copy()]), (r + t).flatten())
    assert np.allclose(oracle.sum_fields([t.copy(), t - 15.0]), (t * 2.0 - 15.0).flatten())
    assert oracle.sum_fields([r.copy(), t.copy()]).shape == (6,)  # arrays are flattened in sum


def test_snow_depth_and_glacier_vectors():
    g = GOLDEN["snow_depth_m"]
    sd, rsn = arr(g["snow_depth"]), arr(g["snow_density"])
    np.testing.assert_allclose(oracle.snow_depth_m(sd, rsn), 1000.0 * sd / rsn)
    k = g["known"]
    np.testing.assert_allclose(oracle.snow_depth_m(arr(k["snow_depth"]), arr(k["snow_density"])), arr(k["expected"]))
    m = GOLDEN["glacier_mask"]
    values, mask = arr(m["snow_depth"]), arr(m["glacier_mask"]).astype(bool)
    expected = np.ma.array(values, mask=mask).filled(np.nan)  # the reference test's own expectation (test_glacier_mask.py:54)
    assert np.allclose(oracle.apply_mask_values(values.flatten(), mask.flatten()).reshape(values.shape), expected, equal_nan=True)


@pytest.mark.parametrize("case,zero_left", [("zero_left_true", True), ("zero_left_false", False)])
def test_accum_to_interval_vectors(case, zero_left):
    g = GOLDEN["accum_to_interval"]
    base, c = arr(g["base"]), g[case]
    fields = [dict(param="tp", values=base * c["accumulated_multiples_of_base"][i], valid_datetime=g["times"][i]) for i in c["given_order"]]
    fields.append(dict(param="t", values=base + 10, valid_datetime=g["times"][0]))
    out = oracle.filter_accum_to_interval(fields, variables=["tp"], zero_left=zero_left)
    tp = sorted((f for f in out if f["param"] == "tp"), key=lambda f: f["valid_datetime"])
    for f, mult in zip(tp, c["expected_multiples"]):
        assert np.allclose(f["values"], base * mult)
    assert [f for f in out if f["param"] == "t"][0]["values"] is fields[-1]["values"]  # non-target variables pass through
