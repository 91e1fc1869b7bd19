"""Multi-input per-point filters (MatchingFieldsFilter family), written like the reference's tests:
R: tests/field_filters/test_cos_sin_from_rad.py, test_cos_sin_mean_wave_direction.py, test_snow_depth_m.py,
test_snow.py, test_sum.py, tests/test_matching.py, tests/test_grouping.py."""

from __future__ import annotations

import json
import os

import numpy as np
import pytest
import torch

from anemoi_transform_amd import native
from anemoi_transform_amd.core import source_registry
from anemoi_transform_amd.fields import fieldlist_from_dicts
from anemoi_transform_amd.filters import create_filter_by_name
from anemoi_transform_amd.filters.multi import MatchingFieldsFilter, MatchingSpec
from anemoi_transform_amd.grouping import GroupByParam
from oracle import oracle

import native_double
from test_filters import collect_fields_by_param, test_source

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))

MD = {"latitudes": [10.0, 0.0, -10.0], "longitudes": [20, 40.0], "valid_datetime": "2018-08-01T09:00:00Z"}

# R: tests/field_filters/test_cos_sin_from_rad.py:19-23 (data literals)
RAD_VALUES = np.array([[2.67687254, 2.59108576], [1.83746659, 1.73104875], [1.1348185, 2.23051268]])
COS_RAD_VALUES = np.array([[-0.89394704, -0.85225947], [-0.26352086, -0.15956740], [0.42229696, -0.61289275]])
SIN_RAD_VALUES = np.array([[0.44817262, 0.52311930], [0.96465370, 0.98718704], [0.90645754, 0.79016611]])
# R: tests/field_filters/test_sum.py:19-27
T_VALUES = np.array([[293.32301331, 284.21559143], [260.53981018, 291.18824768], [279.88941956, 248.87574768]])
Q_VALUES = np.array([[0.00657578, 0.00769957], [0.00147607, 0.01088967], [0.00505508, 0.00044559]])
R_VALUES = np.array([[37.91091442, 79.51638317], [95.61794567, 71.53396130], [70.03982067, 89.69021130]])


@pytest.fixture(params=["double", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request, monkeypatch):
    if request.param == "double":
        native_double.install(monkeypatch)
    return request.param


def mars_test_source(dataset):
    """R: tests/conftest.py:70-80 — fields that expose a MARS namespace."""
    return source_registry.create("testing", dataset=fieldlist_from_dicts(dataset, mars=True))


# ---- wind components ------------------------------------------------------------------------
def _wind_specs(names):
    g = GOLDEN["uv_to_ddff"]
    specs = []
    for level in g["levels"]:
        for param, key in names:
            specs.append({"param": param, "levelist": level, "values": np.array(g[key][str(level)]), **MD})
    return g, specs


def test_uv_to_ddff(engine):
    """R: tests/field_filters/test_uv_to_ddff.py:67-80 — the reference's literal expectations."""
    g, specs = _wind_specs([("u", "u"), ("v", "v")])
    out = collect_fields_by_param(test_source(specs) | create_filter_by_name("uv_to_ddff"))
    assert set(out) == {"ws", "wdir"} and len(out["ws"]) == 2 and len(out["wdir"]) == 2
    for i, level in enumerate(g["levels"]):
        assert np.allclose(out["ws"][i].to_numpy(), np.array(g["ws"][str(level)]))
        assert np.allclose(out["wdir"][i].to_numpy(), np.array(g["wdir"][str(level)]))
        assert out["ws"][i].metadata("levelist") == level and out["wdir"][i].metadata("levelist") == level


def test_ddff_to_uv_and_round_trips(engine):
    """R: test_uv_to_ddff.py:83-130."""
    g, specs = _wind_specs([("ws", "ws"), ("wdir", "wdir")])
    src = test_source(specs)
    out = collect_fields_by_param(src | create_filter_by_name("ddff_to_uv"))
    assert set(out) == {"u", "v"}
    for i, level in enumerate(g["levels"]):
        assert np.allclose(out["u"][i].to_numpy(), np.array(g["u"][str(level)])) and np.allclose(out["v"][i].to_numpy(), np.array(g["v"][str(level)]))
        assert out["u"][i].metadata("levelist") == level
    back = collect_fields_by_param((src | create_filter_by_name("ddff_to_uv")) | create_filter_by_name("uv_to_ddff"))
    assert set(back) == {"ws", "wdir"}
    for i, level in enumerate(g["levels"]):
        assert np.allclose(back["ws"][i].to_numpy(), np.array(g["ws"][str(level)])) and np.allclose(back["wdir"][i].to_numpy(), np.array(g["wdir"][str(level)]))
    _, uv = _wind_specs([("u", "u"), ("v", "v")])
    rt = collect_fields_by_param((test_source(uv) | create_filter_by_name("uv_to_ddff")) | create_filter_by_name("ddff_to_uv"))
    for i, level in enumerate(g["levels"]):
        assert np.allclose(rt["u"][i].to_numpy(), np.array(g["u"][str(level)])) and np.allclose(rt["v"][i].to_numpy(), np.array(g["v"][str(level)]))
    with pytest.raises(AssertionError):
        create_filter_by_name("uv_to_ddff", radians=True)  # R: uv_to_ddff.py:74


# ---- cos / sin ----------------------------------------------------------------------------
def test_cos_sin_from_rad_forward(engine):
    f = create_filter_by_name("cos_sin_from_rad", param="RAD")
    out = collect_fields_by_param(test_source([{"param": "RAD", "values": RAD_VALUES, **MD}]) | f)
    assert set(out) == {"cos_RAD", "sin_RAD"} and len(out["cos_RAD"]) == 1 and len(out["sin_RAD"]) == 1
    np.testing.assert_allclose(out["cos_RAD"][0].to_numpy(), COS_RAD_VALUES)  # the reference's literal expectations
    np.testing.assert_allclose(out["sin_RAD"][0].to_numpy(), SIN_RAD_VALUES)
    np.testing.assert_allclose(out["cos_RAD"][0].to_numpy(), np.cos(RAD_VALUES), rtol=1e-14)


def test_cos_sin_from_rad_reverse_and_round_trip(engine):
    src = test_source([{"param": "cos_RAD", "values": COS_RAD_VALUES, **MD}, {"param": "sin_RAD", "values": SIN_RAD_VALUES, **MD}])
    f = create_filter_by_name("cos_sin_from_rad", param="some_rad", cos_param="cos_RAD", sin_param="sin_RAD").reverse()
    out = collect_fields_by_param(src | f)
    assert set(out) == {"some_rad"} and len(out["some_rad"]) == 1
    np.testing.assert_allclose(out["some_rad"][0].to_numpy(), RAD_VALUES)
    g = create_filter_by_name("cos_sin_from_rad", param="RAD")
    rad = test_source([{"param": "RAD", "values": RAD_VALUES, **MD}])
    mid = collect_fields_by_param(rad | g)
    back = collect_fields_by_param((rad | g) | g.reverse())
    assert set(mid) == {"cos_RAD", "sin_RAD"} and set(back) == {"RAD"}
    np.testing.assert_allclose(back["RAD"][0].to_numpy(), RAD_VALUES)
    assert back["RAD"][0].metadata("valid_datetime") == MD["valid_datetime"]


def test_cos_sin_from_rad_rejects_degrees(engine):
    """R: tests/field_filters/test_cos_sin_from_rad.py:117-130."""
    f = create_filter_by_name("cos_sin_from_rad", param="DEG")
    with pytest.raises(ValueError, match="expected in radians"):
        collect_fields_by_param(test_source([{"param": "DEG", "values": np.rad2deg(RAD_VALUES), **MD}]) | f)


def test_cos_sin_mean_wave_direction(engine):
    mwd = np.rad2deg(RAD_VALUES) * 2.0 % 360.0
    f = create_filter_by_name("cos_sin_mean_wave_direction")
    src = test_source([{"param": "mwd", "values": mwd, **MD}, {"param": "swh", "values": mwd, **MD}])
    out = collect_fields_by_param(src | f)
    assert set(out) == {"cos_mwd", "sin_mwd", "swh"}
    c, s = oracle.cos_sin(mwd, degrees=True)
    np.testing.assert_allclose(out["cos_mwd"][0].to_numpy(), c, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(out["sin_mwd"][0].to_numpy(), s, rtol=1e-13, atol=1e-15)
    back = collect_fields_by_param((src | f) | f.reverse())
    assert set(back) == {"mwd", "swh"}
    got = back["mwd"][0].to_numpy()
    np.testing.assert_allclose(got, mwd, rtol=1e-12)
    assert got.min() >= 0.0 and got.max() < 360.0
    assert f.patch_data_request({"param": ["cos_mwd", "swh"]})["param"] == ["swh", "mwd"]


# ---- snow ------------------------------------------------------------------------------------
def test_snow_depth_and_cover(engine):
    """R: tests/field_filters/test_snow_depth_m.py:24-36 (known values) and snow_cover.py:34-39."""
    sd = np.array([[0.01, 0.02], [0.05, 0.1], [0.1, 0.0]])
    rsn = np.array([[200.0, 250.0], [300.0, 400.0], [500.0, 50.0]])
    specs = [{"param": "sd", "values": sd, **MD}, {"param": "rsn", "values": rsn, **MD}, {"param": "2t", "values": sd, **MD}]
    out = collect_fields_by_param(test_source(specs) | create_filter_by_name("snow_depth_m"))
    assert set(out) == {"sde", "2t"}
    got = out["sde"][0].to_numpy()
    assert np.array_equal(got, oracle.snow_depth_m(sd, rsn))  # one multiply and one divide: bit-exact
    np.testing.assert_allclose(got[2, 0], 0.2)  # 10 cm water equivalent at 500 kg/m3
    assert out["sde"][0].metadata("units") == "m"
    out = collect_fields_by_param(test_source(specs) | create_filter_by_name("snow_cover"))
    assert set(out) == {"snowc", "2t"}
    np.testing.assert_allclose(out["snowc"][0].to_numpy(), oracle.snow_cover(sd, rsn), rtol=1e-13)
    assert out["snowc"][0].metadata("units") == "Fraction"


# ---- vertical velocity ---------------------------------------------------------------------------
def test_w_to_wz_and_back(engine):
    rng = np.random.default_rng(5)
    specs = []
    for level in (500, 850):
        for step in (0, 6):
            specs += [
                {"param": "w", "levelist": level, "step": step, "values": rng.normal(0, 0.5, (3, 2)), **MD},
                {"param": "t", "levelist": level, "step": step, "values": T_VALUES + step, **MD},
                {"param": "q", "levelist": level, "step": step, "values": Q_VALUES, **MD},
            ]
    src = mars_test_source(specs)
    f = create_filter_by_name("w_to_wz")
    out = list(src | f)
    # per group: returned inputs (all three) then wz, groups in order of first appearance
    assert [x.metadata("param") for x in out] == ["w", "t", "q", "wz"] * 4
    for g in range(4):
        w, t, q, wz = out[4 * g: 4 * g + 4]
        level = float(w.metadata("levelist"))
        want = oracle.w_to_wz(w.to_numpy(), t.to_numpy(), q.to_numpy(), level)
        np.testing.assert_allclose(wz.to_numpy(), want, rtol=1e-14)
        assert wz.metadata("levelist") == w.metadata("levelist") and wz.metadata("step") == w.metadata("step")
    # wz_to_w on (wz, t, q) recovers w
    back_in = [x for x in out if x.metadata("param") != "w"]
    from anemoi_transform_amd.fields import FieldList

    back = create_filter_by_name("wz_to_w", return_inputs="none").forward(FieldList(back_in))
    assert [x.metadata("param") for x in back] == ["w"] * 4
    for g in range(4):
        np.testing.assert_allclose(back[g].to_numpy(), out[4 * g].to_numpy(), rtol=1e-6, atol=1e-12)


def test_w_to_wz_needs_a_level(engine):
    specs = [{"param": p, "values": T_VALUES, **MD} for p in ("w", "t", "q")]
    with pytest.raises(TypeError):
        list(test_source(specs) | create_filter_by_name("w_to_wz"))


# ---- sum ------------------------------------------------------------------------------------------
def test_sum_fields(engine):
    """R: tests/field_filters/test_sum.py:48-66."""
    specs = [{"param": "r", "levelist": 850, "values": R_VALUES, **MD}, {"param": "t", "levelist": 850, "values": T_VALUES, **MD},
             {"param": "q", "levelist": 850, "values": Q_VALUES, **MD}]
    out = collect_fields_by_param(mars_test_source(specs) | create_filter_by_name("sum", params=["r", "t"], output="sum"))
    assert set(out) == {"q", "sum"} and len(out["sum"]) == 1
    expected = (R_VALUES + T_VALUES).flatten()
    assert out["sum"][0].to_numpy().shape == expected.shape  # arrays are flattened in sum
    assert np.array_equal(out["sum"][0].to_numpy(), expected)


def test_sum_multilevel_and_errors(engine):
    """R: tests/field_filters/test_sum.py:36-45,80-120."""
    specs = [{"param": "t_850", "levelist": 850, "values": T_VALUES, **MD},
             {"param": "t_500", "levelist": 500, "values": T_VALUES - 15.0, **MD},
             {"param": "r", "levelist": 850, "values": R_VALUES, **MD}]
    src = mars_test_source(specs)
    out = collect_fields_by_param(src | create_filter_by_name("sum", params=["t_850", "t_500"], output="sum", ignore_level=True))
    assert set(out) == {"r", "sum"}
    assert np.array_equal(out["sum"][0].to_numpy(), oracle.sum_fields([T_VALUES, T_VALUES - 15.0]))
    with pytest.raises(ValueError, match="Missing fields"):
        list(src | create_filter_by_name("sum", params=["t_850", "t_500"], output="sum"))
    with pytest.raises(NotImplementedError):
        create_filter_by_name("sum", params=["r"], output="s").backward(None)


# ---- MatchingFieldsFilter contract (R: tests/test_matching.py) -----------------------------------------
def test_matching_filter_contract_and_user_subclass():
    with pytest.raises(TypeError, match="must define a 'MATCHING'"):

        class NoSpec(MatchingFieldsFilter):
            def forward_transform(self, a):
                pass

    with pytest.raises(ValueError, match="missing parameters"):

        class BadSig(MatchingFieldsFilter):
            MATCHING = MatchingSpec(forward=("a", "b"))

            def __init__(self, *, a="a", b="b"):
                super().__init__()

            def forward_transform(self, a):
                pass

    with pytest.raises(NotImplementedError):
        MatchingSpec(select="levelist")
    with pytest.raises(ValueError, match="must subset"):
        MatchingSpec(forward=("a",), return_inputs=("zzz",))
    spec = MatchingSpec(forward="a", backward=("b", "c"), return_inputs="all")
    assert spec.forward == ("a",) and spec.inputs("backward") == ("b", "c")
    assert spec.update_return_inputs("none").inputs("forward") == ()
    assert spec.update_return_inputs(["b"]).inputs("forward") == ("b",)

    class Product(MatchingFieldsFilter):  # numpy inside, as a user of the reference would write it
        MATCHING = MatchingSpec(forward=("a", "b"), return_inputs=("a",))

        def __init__(self, *, a="x", b="y", out="xy"):
            self.a, self.b, self.out = a, b, out
            super().__init__()

        def forward_transform(self, a, b):
            yield self.new_field_from_numpy(a.to_numpy() * b.to_numpy(), template=a, param=self.out)

    specs = [{"param": p, "step": s, "values": np.full((3, 2), v), **MD} for s in (0, 1) for p, v in (("x", 2.0 + s), ("y", 5.0), ("z", 0.0))]
    out = Product()(fieldlist_from_dicts(specs))
    assert [f.metadata("param") for f in out] == ["z", "z", "x", "xy", "x", "xy"]
    assert out[3].to_numpy()[0, 0] == 10.0 and out[5].to_numpy()[0, 0] == 15.0
    # no backward operands declared: nothing matches, every field is passed on (as in the reference)
    assert len(Product().backward(fieldlist_from_dicts(specs))) == len(specs)

    # declaring backward operands without a backward_transform of that signature is refused at class creation
    with pytest.raises(ValueError, match="missing parameters"):

        class OneWay(MatchingFieldsFilter):
            MATCHING = MatchingSpec(forward=("a",), backward=("a",))

            def __init__(self, *, a="x"):
                super().__init__()

            def forward_transform(self, a):
                yield a


def test_group_by_param():
    """R: tests/test_grouping.py:57-87."""
    specs = [{"param": p, "levelist": lev, "values": np.zeros((3, 2)), **MD} for lev in (500, 850) for p in ("u", "v", "t")]
    fl = fieldlist_from_dicts(specs, mars=True)
    others = []
    groups = list(GroupByParam(["u", "v"]).iterate(fl, other=others.append))
    assert len(groups) == 2 and [f.metadata("param") for f in others] == ["t", "t"]
    for u, v in groups:
        assert (u.metadata("param"), v.metadata("param")) == ("u", "v") and u.metadata("levelist") == v.metadata("levelist")
    with pytest.raises(ValueError, match="Lost field"):
        list(GroupByParam(["u", "v"]).iterate(fl))
    with pytest.raises(ValueError, match="Missing component"):
        list(GroupByParam(["u", "v"]).iterate(fl[:1], other=others.append))
    with pytest.raises(ValueError, match="Duplicate component"):
        list(GroupByParam(["u"]).iterate(fieldlist_from_dicts([specs[0], specs[0]], mars=True)))
    assert GroupByParam([["u", "v"], "t"]).params == ["u", "v", "t"]


# ---- the kernel, directly, against the oracle (GPU) ---------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("tdtype,np_dtype,rtol", [(torch.float64, np.float64, 1e-13), (torch.float32, np.float32, 2e-6)])
@pytest.mark.parametrize("layout", [native.COLUMNS, native.FIELDS])
def test_combine_kernel_vs_oracle(dev, tdtype, np_dtype, rtol, layout):
    from anemoi_transform_amd.stack import Stack

    rng = np.random.default_rng(21)
    n_lev, n_pts = 7, 4099
    a = rng.uniform(0.0, 0.3, (n_lev, n_pts)).astype(np_dtype)
    b = rng.uniform(50.0, 600.0, (n_lev, n_pts)).astype(np_dtype)
    c = rng.uniform(0.0, 0.02, (n_lev, n_pts)).astype(np_dtype)
    ang = rng.uniform(-6.0, 6.0, (n_lev, n_pts)).astype(np_dtype)
    levels = np.array([10.0, 100.0, 250.0, 500.0, 700.0, 850.0, 1000.0])

    def run(op, ins, n_out, flags=0, with_levels=False):
        stacks = [Stack.from_fields(x, dev=dev, layout=layout) for x in ins]
        outs = [stacks[0].new_like(zero=False) for _ in range(n_out)]
        lp = torch.from_numpy(levels).to(dev) if with_levels else None
        native.combine_stack(op, [s.data for s in stacks], [o.data for o in outs], n_pts=n_pts, n_lev=n_lev, pitch=stacks[0].pitch,
                             layout=layout, level_param=lp, flags=flags)
        return [o.numpy() for o in outs]

    assert np.array_equal(run(native.COMB_SNOW_DEPTH_M, [a, b], 1)[0], oracle.snow_depth_m(a, b))
    np.testing.assert_allclose(run(native.COMB_SNOW_COVER, [a, b], 1)[0], oracle.snow_cover(a, b), rtol=rtol, atol=1e-7)
    for deg in (False, True):
        x = np.rad2deg(ang).astype(np_dtype) if deg else ang
        co, si = run(native.COMB_COS_SIN, [x], 2, flags=native.COMB_DEGREES if deg else 0)
        wc, ws = oracle.cos_sin(x, deg)
        np.testing.assert_allclose(co, wc, rtol=rtol, atol=1e-6 if np_dtype == np.float32 else 1e-15)
        np.testing.assert_allclose(si, ws, rtol=rtol, atol=1e-6 if np_dtype == np.float32 else 1e-15)
        back = run(native.COMB_ATAN2, [wc.astype(np_dtype), ws.astype(np_dtype)], 1, flags=native.COMB_DEGREES if deg else 0)[0]
        np.testing.assert_allclose(back, oracle.direction_from_cos_sin(wc.astype(np_dtype), ws.astype(np_dtype), deg), rtol=rtol, atol=1e-4)
    t = (250.0 + 40.0 * rng.random((n_lev, n_pts))).astype(np_dtype)
    w = rng.normal(0, 0.5, (n_lev, n_pts)).astype(np_dtype)
    want = np.stack([oracle.w_to_wz(w[l], t[l], c[l], np_dtype(levels[l])) for l in range(n_lev)])
    np.testing.assert_allclose(run(native.COMB_W_TO_WZ, [w, t, c], 1, with_levels=True)[0], want, rtol=rtol)
    want = np.stack([oracle.wz_to_w(w[l], t[l], c[l], np_dtype(levels[l])) for l in range(n_lev)])
    np.testing.assert_allclose(run(native.COMB_WZ_TO_W, [w, t, c], 1, with_levels=True)[0], want, rtol=rtol)
    terms = [a, b, c, t, w]
    assert np.array_equal(run(native.COMB_SUM, terms, 1)[0], np.stack([oracle.sum_fields([x[l] for x in terms]) for l in range(n_lev)]))
    # wind components <-> speed / direction (R: uv_to_ddff.py:93-97, 121-125)
    u, v = rng.normal(0, 8, (n_lev, n_pts)).astype(np_dtype), rng.normal(0, 8, (n_lev, n_pts)).astype(np_dtype)
    u[0, :4], v[0, :4] = [0.0, 0.0, -3.0, 3.0], [5.0, -5.0, 0.0, 0.0]  # from the south / north / east / west: 180, 0, 90, 270 degrees
    ws, wd = run(native.COMB_XY_TO_POLAR, [u, v], 2)
    want_ws, want_wd = oracle.xy_to_polar(u, v)
    np.testing.assert_allclose(ws, want_ws, rtol=rtol)
    np.testing.assert_allclose(wd, want_wd, rtol=rtol, atol=1e-9 if np_dtype == np.float64 else 1e-3)
    assert list(wd[0, :4]) == [180.0, 0.0, 90.0, 270.0] and wd.min() >= 0.0 and wd.max() < 360.0
    bu, bv = run(native.COMB_POLAR_TO_XY, [want_ws.astype(np_dtype), want_wd.astype(np_dtype)], 2)
    wu, wv = oracle.polar_to_xy(want_ws.astype(np_dtype), want_wd.astype(np_dtype))
    np.testing.assert_allclose(bu, wu, rtol=rtol, atol=1e-12 if np_dtype == np.float64 else 1e-5)
    np.testing.assert_allclose(bv, wv, rtol=rtol, atol=1e-12 if np_dtype == np.float64 else 1e-5)
    np.testing.assert_allclose(bu, u, rtol=1e-9 if np_dtype == np.float64 else 2e-4, atol=1e-9 if np_dtype == np.float64 else 2e-4)
    with pytest.raises(ValueError):
        run(native.COMB_SNOW_COVER, [a], 1)


@pytest.mark.gpu
def test_snow_cover_float64_tanh_within_ulps_of_numpy(dev):
    """float64 snow_cover evaluates tanh as expm1(2x) / (expm1(2x) + 2) (atx_combine.hip) instead of the device library's tanh: on
    the arguments where tanh decides the value — (0, 2.65] — the result stays within 4 ulp of numpy's."""
    from anemoi_transform_amd.stack import Stack

    rng = np.random.default_rng(9)
    n = 1 << 18
    arg = np.concatenate([rng.uniform(0.0, 2.65, n // 2), 10.0 ** rng.uniform(-12, 0.4, n // 2)])
    # next to where the reduction of expm1(2 arg) switches (2 arg log2(e) at a half-integer: k = 0 .. 7) and where its r is next to 0
    edges = np.concatenate([(np.arange(0, 8) + 0.5) * np.log(2.0) / 2.0, np.arange(1, 8) * np.log(2.0) / 2.0])
    edges = edges[edges < 2.65]
    arg[:edges.size * 41] = (edges[:, None] * (1.0 + np.arange(-20, 21)[None, :] * 2.0 ** -51)).reshape(-1)
    rsn = rng.uniform(100.0, 400.0, n)
    sd = arg * rsn * rsn / 4.0e6  # 4000 * (1000 * sd / rsn) / clip(rsn, 100, 400) == arg up to rounding
    want = oracle.snow_cover(sd[None, :].copy(), rsn[None, :].copy())[0]
    a, b = Stack.from_fields(sd[None, :], dev=dev), Stack.from_fields(rsn[None, :], dev=dev)
    out = a.new_like(zero=False)
    native.combine_stack(native.COMB_SNOW_COVER, [a.data, b.data], [out.data], n_pts=n, n_lev=1, pitch=a.pitch, layout=native.COLUMNS)
    got = out.numpy()[0]
    live = (want > 0) & (want < 1)
    assert live.sum() > n // 2
    assert float(np.max(np.abs(got[live] - want[live]) / np.spacing(want[live]))) <= 4.0
    assert np.array_equal(got[~live], want[~live])


@pytest.mark.gpu
def test_atan2_float64_within_ulps_of_numpy(dev):
    """The float64 direction-from-(cos, sin) operator (the backward transform of the cos / sin filters, R: cos_sin_from_rad.py:100,
    cos_sin_mean_wave_direction.py:97-99: np.arctan2): at most 2 ulp from numpy (each side within 1 ulp of the true value) on the unit
    circle, on random magnitudes, next to the axes, the diagonals and the edges of fdlibm's reduction intervals, for magnitudes up to
    1e+-300; IEEE's values — signs of zero included — for zeros, infinities and NaN.  (The operator runs the device library's atan2: an
    own routine — one quotient for ratio and reduction, fdlibm's coefficients on scalar-register Horner steps, <= 1 ulp from numpy on
    these same cases — was written in round 5, measured at the SAME speed, 3.74 against 3.73 ms with 20 more registers, and not adopted:
    tools/experiments/atan2_own.patch, profiles/r05_atan2_experiment.log.)"""
    from anemoi_transform_amd.stack import Stack

    rng = np.random.default_rng(11)
    n = 1 << 18
    th = rng.uniform(-np.pi, np.pi, n)
    sgn = lambda: rng.choice([-1.0, 1.0], n)  # noqa: E731
    edge = np.concatenate([0.4375 * (1 + rng.uniform(-1e-12, 1e-12, n // 2)), 0.6875 * (1 + rng.uniform(-1e-12, 1e-12, n - n // 2))])
    mag = 10.0 ** rng.uniform(-5, 5, n)
    cases = {  # (y, x)
        "unit circle": (np.sin(th), np.cos(th)),
        "random magnitudes": (rng.standard_normal(n) * mag, rng.standard_normal(n) * 10.0 ** rng.uniform(-5, 5, n)),
        "next to the axes": (rng.standard_normal(n) * 1e-9, sgn() * rng.uniform(0.5, 2.0, n)),
        "next to the y axis": (sgn() * rng.uniform(0.5, 2.0, n), rng.standard_normal(n) * 1e-9),
        "next to the diagonals": ((1 + rng.uniform(-1e-9, 1e-9, n)) * sgn(), (1 + rng.uniform(-1e-9, 1e-9, n)) * sgn()),
        "interval edges": (edge * sgn(), sgn()),
        "interval edges, swapped": (sgn(), edge * sgn()),
        "enormous": (rng.standard_normal(n) * 1e300, rng.standard_normal(n) * 1e300),
        "minute": (rng.standard_normal(n) * 1e-300, rng.standard_normal(n) * 1e-300),
        "far apart": (rng.standard_normal(n) * 1e-200, rng.standard_normal(n) * 1e200),
    }

    def run(y, x, flags=0):
        c, s_ = Stack.from_fields(x[None, :], dev=dev), Stack.from_fields(y[None, :], dev=dev)  # operands of the operator: (cos, sin)
        out = c.new_like(zero=False)
        native.combine_stack(native.COMB_ATAN2, [c.data, s_.data], [out.data], n_pts=x.size, n_lev=1, pitch=c.pitch, layout=native.COLUMNS, flags=flags)
        return out.numpy()[0]

    for name, (y, x) in cases.items():
        got, want = run(y, x), np.arctan2(y, x)
        tiny = np.abs(want) < 2.3e-308  # a subnormal result has no ulp of its own to count in: one subnormal step
        err = np.abs(got - want) / np.where(tiny, 5e-324, np.spacing(np.abs(want)))
        assert float(err.max()) <= 2.0, (name, float(err.max()))
        assert np.array_equal(np.signbit(got), np.signbit(want)), name
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 1.7976931348623157e308, 2.2250738585072014e-308])
    Y, X = (a.reshape(-1).copy() for a in np.meshgrid(special, special))
    got, want = run(Y, X), np.arctan2(Y, X)
    assert np.array_equal(got, want, equal_nan=True), [(y, x, g, w) for y, x, g, w in zip(Y, X, got, want) if not (g == w or (g != g and w != w))]
    assert np.array_equal(np.signbit(got[~np.isnan(want)]), np.signbit(want[~np.isnan(want)]))
    # in degrees, wrapped to [0, 360) as the wave-direction filter asks (oracle.direction_from_cos_sin): one more multiply and the wrap
    y, x = cases["unit circle"]
    got = run(y, x, flags=native.COMB_DEGREES)
    want = oracle.direction_from_cos_sin(x, y, degrees=True)
    assert got.min() >= 0.0 and got.max() < 360.0
    assert float(np.max(np.abs(got - want) / np.spacing(np.maximum(np.abs(want), 1.0)))) <= 3.0


@pytest.mark.gpu
def test_fast_sincos_float64_within_ulps_of_numpy(dev):
    """The float64 cos+sin operator reduces moderate arguments itself (atx_combine.hip: sincos_moderate) instead of calling the
    device library's general routine: at most 2 ulp from numpy (each side is within 1 ulp of the true value) on the ranges the
    filters see, exact signs and zeros next to multiples of pi/2, the library routine (same bound) beyond 1e5, NaN / inf -> NaN."""
    from anemoi_transform_amd.stack import Stack

    rng = np.random.default_rng(5)
    n = 1 << 18
    near = (np.arange(-200, 200)[:, None] * (np.pi / 2) + rng.uniform(-1e-9, 1e-9, (400, n // 400))).reshape(-1)
    cases = {
        "[-2 pi, 2 pi]": rng.uniform(-2 * np.pi, 2 * np.pi, n),
        "[-1e5, 1e5]": rng.uniform(-1e5, 1e5, n),
        "next to multiples of pi/2": np.resize(near, n),
        "tiny": rng.uniform(-1e-8, 1e-8, n),
        "beyond the fast path": rng.uniform(1e5, 1e15, n) * rng.choice([-1.0, 1.0], n),
    }

    def run(x, flags=0):
        st = Stack.from_fields(x[None, :], dev=dev)
        co, si = st.new_like(zero=False), st.new_like(zero=False)
        native.combine_stack(native.COMB_COS_SIN, [st.data], [co.data, si.data], n_pts=x.size, n_lev=1, pitch=st.pitch, layout=native.COLUMNS, flags=flags)
        return co.numpy()[0], si.numpy()[0]

    def ulps(got, want):
        return float(np.max(np.abs(got - want) / np.spacing(np.abs(want))))

    for name, x in cases.items():
        co, si = run(x)
        assert ulps(co, np.cos(x)) <= 2.0 and ulps(si, np.sin(x)) <= 2.0, (name, ulps(co, np.cos(x)), ulps(si, np.sin(x)))
    # degrees: the statement is cos(deg2rad(x)) (R: cos_sin_mean_wave_direction.py:72-76)
    deg = rng.uniform(0.0, 360.0, n)
    co, si = run(deg, flags=native.COMB_DEGREES)
    assert ulps(co, np.cos(np.deg2rad(deg))) <= 2.0 and ulps(si, np.sin(np.deg2rad(deg))) <= 2.0
    special = np.array([0.0, -0.0, np.pi / 2, np.pi, -np.pi, 2 * np.pi, np.inf, -np.inf, np.nan, 1e5, -1e5, 99999.999], dtype=np.float64)
    co, si = run(special)
    with np.errstate(invalid="ignore"):
        wc, ws = np.cos(special), np.sin(special)
    assert np.array_equal(np.isnan(co), np.isnan(wc)) and np.array_equal(np.isnan(si), np.isnan(ws))
    ok = ~np.isnan(wc)
    assert ulps(co[ok], wc[ok]) <= 2.0 and ulps(si[ok], ws[ok]) <= 2.0
    assert si[0] == 0.0 and np.signbit(si[1]) and co[0] == 1.0  # sin(-0.0) = -0.0


@pytest.mark.gpu
@pytest.mark.parametrize("tdtype,np_dtype", [(torch.float64, np.float64), (torch.float32, np.float32)])
def test_snow_cover_shortcuts_give_the_statements_values(dev, tdtype, np_dtype):
    """R: filters/fields/snow_cover.py:34-39.  The kernel skips tanh for deep snow (argument beyond atanh(0.99): the statement's
    threshold makes the result exactly 1.0) and for bare ground (tanh(+-0) = +-0): around both decisions, and for NaN / inf /
    negative inputs, the result must be the statement's — exactly where no tanh is involved, to ocml-vs-libm rounding where it is."""
    from anemoi_transform_amd.stack import Stack

    rsn = np.array([50.0, 100.0, 250.0, 400.0, 900.0])
    clipped = np.clip(rsn, 100, 400)
    # snow depths placing 4000 * (1000 * sd / rsn) / clip(rsn) at chosen arguments, the decision points included
    args = np.array([0.0, -0.0, 1e-300, 1e-8, 0.5, 2.0, 2.6, 2.6466, 2.6467, 2.649999, 2.65, 2.650001, 2.7, 5.0, 19.0, 40.0, 1e6, -1e-3, -3.0,
                     np.inf, -np.inf, np.nan])
    sd = (args[:, None] * clipped[None, :] * rsn[None, :] / 4.0e6).astype(np_dtype)
    den = np.broadcast_to(rsn[None, :], sd.shape).astype(np_dtype)
    n = sd.size
    a = np.tile(sd.reshape(1, -1), (3, 1))
    b = np.tile(den.reshape(1, -1), (3, 1))
    b[2, ::7] = np.nan  # NaN density too
    want = oracle.snow_cover(a.copy(), b.copy())
    for layout in (native.COLUMNS, native.FIELDS):
        sa, sb = Stack.from_fields(a, dev=dev, layout=layout), Stack.from_fields(b, dev=dev, layout=layout)
        out = sa.new_like(zero=False)
        native.combine_stack(native.COMB_SNOW_COVER, [sa.data, sb.data], [out.data], n_pts=n, n_lev=3, pitch=sa.pitch, layout=layout)
        got = out.numpy()
        assert np.array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        exact = ok & ((want == 1.0) | (want == 0.0))  # saturated, bare or clipped: no tanh value survives
        assert np.array_equal(got[exact], want[exact]) and np.array_equal(np.signbit(got[exact]), np.signbit(want[exact]))
        np.testing.assert_allclose(got[ok], want[ok], rtol=1e-13 if np_dtype == np.float64 else 2e-6)
        assert (want[ok] == 1.0).sum() > n // 4 and ((want[ok] > 0) & (want[ok] < 0.99)).sum() > n // 8  # both sides of the decision are exercised


def test_accum_to_interval(engine):
    """R: tests/field_filters/test_accum_to_interval.py — differencing within (param, level) groups by valid time."""
    rng = np.random.default_rng(8)
    times = ["2020-01-01T06:00:00", "2020-01-01T00:00:00", "2020-01-01T12:00:00", "2020-01-01T18:00:00"]
    acc = {t: np.cumsum(rng.random((3, 2)), axis=0) * (i + 1) for i, t in enumerate(sorted(times))}
    specs = []
    for t in times:  # deliberately not in time order
        specs.append({"param": "tp", "values": acc[t], "latitudes": MD["latitudes"], "longitudes": MD["longitudes"], "valid_datetime": t})
        specs.append({"param": "2t", "values": acc[t] + 270.0, "latitudes": MD["latitudes"], "longitudes": MD["longitudes"], "valid_datetime": t})
    for zero_left in (True, False):
        out = list(test_source(specs) | create_filter_by_name("accum_to_interval", variables=["tp"], zero_left=zero_left))
        want = oracle.filter_accum_to_interval([dict(s) for s in specs], variables=["tp"], zero_left=zero_left)
        assert [(f.metadata("param"), f.metadata("valid_datetime")) for f in out] == [(w["param"], w["valid_datetime"]) for w in want]
        for f, w in zip(out, want):
            assert np.array_equal(f.to_numpy(), np.asarray(w["values"]))
    tp = [f for f in out if f.metadata("param") == "tp"]
    assert [f.metadata("valid_datetime") for f in tp] == sorted(times)
