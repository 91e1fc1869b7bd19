"""Filter-level parity tests, written like the reference's own field-filter tests.

Each test builds its input with the reference tests' fixture form (list of dicts ->
FieldList), runs ``source | filter`` through the registry, and checks the result
against the oracle / the reference's literal expectations.

Two engines:
  * ``double`` — host logic on the CPU box: ``native`` is monkeypatched with the
    oracle-backed test double (tests/native_double.py);
  * ``hip``    — the real thing on an MI355X through the C ABI (``-m gpu``).
"""

from __future__ import annotations

import json
import os

import numpy as np
import pytest

from anemoi_transform_amd.core import Source, source_registry
from anemoi_transform_amd.fields import fieldlist_from_dicts
from anemoi_transform_amd.filters import create_filter, create_filter_by_name, filter_registry
from oracle import oracle

import native_double

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))


@pytest.fixture(params=["double", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request, monkeypatch):
    if request.param == "double":
        native_double.install(monkeypatch)
    else:
        import torch

        assert torch.cuda.is_available()
        from anemoi_transform_amd import native

        native.load()
    return request.param


if not source_registry.is_registered("testing"):

    @source_registry.register("testing")
    class TestingSource(Source):
        """R: tests/conftest.py:40-50."""

        __test__ = False

        def __init__(self, *, dataset) -> None:
            assert dataset is not None, "Dataset cannot be None"
            self.ds = dataset

        def forward(self, *args, **kwargs):
            return self.ds


def test_source(dataset):
    """R: tests/conftest.py:53-67 (list-of-dicts branch)."""
    return source_registry.create("testing", dataset=fieldlist_from_dicts(dataset))


test_source.__test__ = False


def collect_fields_by_param(pipeline):
    """R: tests/utils/__init__.py:17-22."""
    fields = {}
    for field in pipeline:
        fields.setdefault(field.metadata("param"), []).append(field)
    return fields


def nan_rows(x):
    return np.array([[np.nan if v is None else v for v in row] for row in x], dtype=float)


# =================================================================================
# apply_mask (R: tests/field_filters/test_apply_mask.py)
# =================================================================================
AM = GOLDEN["apply_mask"]


@pytest.fixture
def am_source():
    specs = [{"param": p, "values": np.array(v, float), **AM["metadata"]} for p, v in AM["data_values"].items()]
    return test_source(specs)


@pytest.fixture
def mask_files(tmp_path):
    paths = {}
    for name, values in AM["mask_values"].items():
        path = tmp_path / f"{name}.npy"
        np.save(path, np.array(values).flatten())
        paths[name] = str(path)
    return paths


def test_apply_mask_fails_without_arguments(mask_files):
    with pytest.raises(ValueError):
        create_filter_by_name("apply_mask", path=mask_files["all_zeros"])


@pytest.mark.parametrize("threshold_options", AM["threshold_options"], ids=str)
@pytest.mark.parametrize("rename", [None, "renamed"])
@pytest.mark.parametrize("mask_name", list(AM["mask_values"]))
def test_apply_mask(engine, am_source, mask_files, mask_name, rename, threshold_options):
    apply_mask = create_filter_by_name("apply_mask", path=mask_files[mask_name], rename=rename, **threshold_options)
    pipeline = am_source | apply_mask
    input_fields = collect_fields_by_param(am_source)
    output_fields = collect_fields_by_param(pipeline)

    expected_mask = np.array(AM["mask_values"][mask_name]).flatten()
    if "mask_value" in threshold_options:
        expected_mask = expected_mask == threshold_options["mask_value"]
    else:
        op = {"<": np.less, ">": np.greater}[threshold_options["threshold_operator"]]
        expected_mask = op(expected_mask, threshold_options["threshold"])
    for param in AM["data_values"]:
        result_param = f"{param}_{rename}" if rename else param
        assert result_param in output_fields
        for input_field, output_field in zip(input_fields[param], output_fields[result_param]):
            expected = input_field.to_numpy(flatten=True).copy()
            expected[expected_mask] = np.nan
            result = output_field.to_numpy(flatten=True)
            assert np.array_equal(expected, result, equal_nan=True)
            assert np.sum(np.isnan(result)) == np.sum(expected_mask)
            # shape of the unflattened field is kept
            assert output_field.to_numpy().shape == (3, 2)


def test_apply_mask_only_single_param(engine, am_source, mask_files):
    apply_mask = create_filter_by_name("apply_mask", path=mask_files["mixed_floats"], threshold=0.5, threshold_operator=">", param="t")
    inputs = collect_fields_by_param(am_source)
    outputs = collect_fields_by_param(am_source | apply_mask)
    expected_mask = np.array(AM["mask_values"]["mixed_floats"]).flatten() > 0.5
    for param in AM["data_values"]:
        for i, o in zip(inputs[param], outputs[param]):
            if param == "t":
                expected = i.to_numpy(flatten=True).copy()
                expected[expected_mask] = np.nan
                assert np.array_equal(expected, o.to_numpy(flatten=True), equal_nan=True)
            else:
                assert o is i  # unselected fields pass through by identity (R: filter.py:193-194)


@pytest.mark.parametrize("op", [">", "<", "==", "!=", ">=", "<=", "gt", "lt", "eq", "ne", "ge", "le"])
def test_apply_mask_all_operator_spellings(engine, am_source, mask_files, op):
    """R: apply_mask.py:23-36."""
    f = create_filter_by_name("apply_mask", path=mask_files["mixed_floats"], threshold=0.5, threshold_operator=op)
    out = collect_fields_by_param(am_source | f)
    m = oracle.compute_mask(np.array(AM["mask_values"]["mixed_floats"]).flatten(), threshold=0.5, threshold_operator=op)
    assert np.array_equal(np.isnan(out["t"][0].to_numpy(flatten=True)), m)


def test_apply_mask_invalid_operator(mask_files):
    with pytest.raises(ValueError, match="Invalid threshold operator"):
        create_filter_by_name("apply_mask", path=mask_files["all_ones"], threshold=0.5, threshold_operator="~")


# =================================================================================
# apply_mask from a field (R: tests/field_filters/test_apply_mask_from_field.py)
# =================================================================================
AF = GOLDEN["apply_mask_from_field"]


@pytest.fixture
def af_source():
    return test_source([{"param": p, "values": np.array(v, float), **AF["metadata"]} for p, v in AF["data_values"].items()])


@pytest.mark.parametrize("case", AF["cases"], ids=lambda c: str(c["options"]))
def test_apply_mask_from_field(engine, af_source, case):
    f = create_filter_by_name("apply_mask", **case["options"])
    out = collect_fields_by_param(af_source | f)
    lsm = np.array(AF["lsm"]).flatten()
    expected_mask = (lsm == 0) if case["mask_rule"] == "lsm == 0" else (lsm < 0.5)
    assert ("lsm" in out) == case["lsm_in_output"]
    for param in case["masked"]:
        expected = np.array(AF["data_values"][param], float).flatten()
        expected[expected_mask] = np.nan
        for field in out[param]:
            assert np.array_equal(field.to_numpy(flatten=True), expected, equal_nan=True)
    for param in case.get("unchanged", []):
        assert np.array_equal(out[param][0].to_numpy(flatten=True), np.array(AF["data_values"][param], float).flatten())
    if case["lsm_in_output"]:
        assert len(out["lsm"]) == 1
        assert np.array_equal(out["lsm"][0].to_numpy(flatten=True), lsm)


def test_apply_mask_from_field_missing_param(engine, af_source):
    f = create_filter_by_name("apply_mask", mask_param="nonexistent", mask_value=0)
    with pytest.raises(ValueError, match="not found in input data"):
        list(af_source | f)


def test_apply_mask_fails_without_path_or_mask_param():
    with pytest.raises(ValueError, match="Exactly one of `path` or `mask_param`"):
        create_filter_by_name("apply_mask_fields", mask_value=0)


def test_apply_mask_fails_with_both_path_and_mask_param():
    with pytest.raises(ValueError, match="Exactly one of `path` or `mask_param`"):
        create_filter_by_name("apply_mask", path="some_file.npy", mask_param="lsm", mask_value=0)


# =================================================================================
# remove_nans (R: tests/field_filters/test_remove_nans.py)
# =================================================================================
RN = GOLDEN["remove_nans"]


@pytest.fixture
def rn_source():
    return test_source([{"param": "t", "step": i, "values": nan_rows(v), **RN["metadata"]} for i, v in enumerate(RN["input_values"])])


def test_remove_nans(engine, rn_source):
    remove_nans = create_filter_by_name("remove_nans")
    inputs = collect_fields_by_param(rn_source)
    outputs = collect_fields_by_param(rn_source | remove_nans)
    assert set(inputs) == {"t"} == set(outputs)
    assert len(inputs["t"]) == len(outputs["t"])
    for i, (fi, fo) in enumerate(zip(inputs["t"], outputs["t"])):
        assert np.array_equal(fi.to_numpy(flatten=True), nan_rows(RN["input_values"][i]).flatten(), equal_nan=True)
        expected = np.array([np.nan if v is None else v for v in RN["expected_values"][i]])
        assert np.array_equal(fo.to_numpy(flatten=True), expected, equal_nan=True)
        lats, lons = fo.grid_points()
        assert np.array_equal(lats, RN["expected_latitudes"])
        assert np.array_equal(lons, RN["expected_longitudes"])
        assert fo.shape == (5,)
        assert fo.metadata("step") == i  # everything else is inherited from the input field


def test_remove_nans_field_on_another_grid_raises_numpys_error(engine):
    """R: remove_nans.py:113 `data[self._mask]`: the mask comes from the FIRST field; a later field on another grid fails numpy's
    boolean indexing with IndexError — here before any launch."""
    from anemoi_transform_amd.grids import lookup

    with pytest.raises(IndexError, match="boolean index did not match"):
        np.zeros(7)[np.ones(5, dtype=bool)]
    specs = synthetic_fields(lookup("o16"), 2, nan_frac=0.1) + synthetic_fields(lookup("o8"), 1)
    with pytest.raises(IndexError, match="boolean index did not match"):
        list(test_source(specs) | create_filter_by_name("remove_nans"))


def test_filters_on_fields_of_zero_points(engine):
    """A first field that is NaN everywhere leaves `remove_nans` with fields of ZERO points (R: remove_nans.py:101-116: `data[mask]` of an
    all-False mask); the filters behind it must take such fields — numpy does — instead of tripping over empty device buffers."""
    from anemoi_transform_amd.grids import lookup

    g = lookup("o8")
    n = len(g["latitudes"])
    specs = [{"param": p, "levelist": 1, "values": np.full(n, np.nan) if i == 0 else 250.0 + np.arange(n, dtype=np.float64),
              "latitudes": g["latitudes"], "longitudes": g["longitudes"]} for i, p in enumerate(["t", "q", "orog"])]
    pipeline = (test_source(specs) | create_filter_by_name("remove_nans") | create_filter_by_name("rescale", scale=2.0, offset=1.0, param="q")
                | create_filter_by_name("orog_to_z") | create_filter_by_name("clip", param="t", minimum=0.0))
    out = list(pipeline)
    assert [f.metadata("param") for f in out] == ["t", "q", "z"]
    for f in out:
        assert f.to_numpy(flatten=True).shape == (0,) and f.grid_points()[0].shape == (0,)
    # and a regrid of such fields through an (empty) index list
    again = list(test_source(specs) | create_filter_by_name("remove_nans") | create_filter_by_name("regrid", mask=np.zeros(0, dtype=np.int64)))
    assert len(again) == 3 and all(f.to_numpy().shape == (0,) for f in again)


def test_remove_nans_invalid_method():
    with pytest.raises(AssertionError, match="Method invalid_method not implemented"):
        create_filter_by_name("remove_nans", method="invalid_method")


def test_remove_nans_with_check():
    with pytest.raises(AssertionError, match="Check not implemented"):
        create_filter_by_name("remove_nans", check=True)


def test_remove_nans_param(engine):
    """R: tests/field_filters/test_remove_nans.py:48-72,113-130."""
    specs = [{"param": "t", "step": i, "values": nan_rows(v), **RN["metadata"]} for i, v in enumerate(RN["input_values"])]
    specs += [{"param": "a", "step": i, "values": nan_rows(v), **RN["metadata"]} for i, v in enumerate(RN["input_values"][::-1])]
    source = test_source(specs)
    out = {}
    for param in ["a", "t", None]:
        out[param] = collect_fields_by_param(source | create_filter_by_name("remove_nans", param=param))
    for i in range(3):
        assert out[None]["a"][i].shape == out[None]["t"][i].shape
    assert out["a"]["a"][0].shape != out["t"]["a"][0].shape
    assert out["a"]["t"][0].shape != out["t"]["t"][0].shape
    assert out["t"]["t"][0].shape == out[None]["t"][0].shape
    want = oracle.filter_remove_nans(
        [dict(param=s["param"], values=s["values"], latitudes=np.repeat(RN["metadata"]["latitudes"], 3),
              longitudes=np.tile(RN["metadata"]["longitudes"], 3)) for s in specs], param="a")
    for f, w in zip(source | create_filter_by_name("remove_nans", param="a"), want):
        assert np.array_equal(f.to_numpy(flatten=True), w["values"], equal_nan=True)


def test_remove_nans_param_missing(engine, rn_source):
    with pytest.raises(ValueError, match="not found"):
        list(rn_source | create_filter_by_name("remove_nans", param="zz"))


# =================================================================================
# orog_to_z (R: tests/field_filters/test_orog_to_z.py)
# =================================================================================
OZ = GOLDEN["orog_to_z"]
OROG = np.array(OZ["orog"])
Z = OROG * GOLDEN["constants"]["g"]


def test_orog_to_z(engine):
    source = test_source([{"param": "orog", "values": OROG, **OZ["metadata"]}])
    out = collect_fields_by_param(source | create_filter_by_name("orog_to_z"))
    assert set(out) == {"z"} and len(out["z"]) == 1
    assert np.allclose(out["z"][0].to_numpy(), Z)
    assert np.array_equal(out["z"][0].to_numpy(), oracle.orog_to_z(OROG))  # bit-exact vs the reference statement


def test_orog_to_z_round_trip(engine):
    source = test_source([{"param": "orog", "values": OROG, **OZ["metadata"]}])
    z_source = source | create_filter_by_name("orog_to_z")
    pipeline = z_source | create_filter_by_name("z_to_orog")
    mid = collect_fields_by_param(z_source)
    out = collect_fields_by_param(pipeline)
    assert set(mid) == {"z"} and set(out) == {"orog"}
    assert np.allclose(out["orog"][0].to_numpy(), OROG)
    assert np.array_equal(out["orog"][0].to_numpy(), oracle.z_to_orog(oracle.orog_to_z(OROG)))
    assert out["orog"][0].metadata("valid_datetime") == OZ["metadata"]["valid_datetime"]
    assert np.array_equal(out["orog"][0].grid_points()[0], source.ds[0].grid_points()[0])


def test_z_to_orog(engine):
    source = test_source([{"param": "z", "values": Z, **OZ["metadata"]}])
    out = collect_fields_by_param(source | create_filter_by_name("z_to_orog"))
    assert set(out) == {"orog"}
    assert np.allclose(out["orog"][0].to_numpy(), OROG)


def test_orog_custom_names_and_passthrough(engine):
    source = test_source([{"param": "h", "values": OROG, **OZ["metadata"]}, {"param": "2t", "values": OROG + 1, **OZ["metadata"]}])
    f = create_filter(None, {"orog_to_z_fields": {"orography": "h", "geopotential": "gh"}})
    out = collect_fields_by_param(source | f)
    assert set(out) == {"gh", "2t"}
    assert out["2t"][0] is source.ds[1]
    back = collect_fields_by_param((source | f) | f.reverse())
    assert np.allclose(back["h"][0].to_numpy(), OROG)


def test_orog_patch_data_request():
    """R: orog_to_z.py:80-94."""
    f = create_filter_by_name("orog_to_z_fields")
    assert f.patch_data_request({"param": ["z", "t"], "levtype": "pl"})["param"] == ["orog", "t"]
    assert f.patch_data_request({"param": ["orog"], "levelist": [500]})["param"] == ["z"]
    assert f.patch_data_request({"param": ["z"], "levtype": "sfc"})["param"] == ["z"]
    with pytest.raises(ValueError):
        f.patch_data_request({"param": ["z", "orog"]})


# =================================================================================
# lnsp_to_sp, impute_nans, clip, rescale, convert, glacier_mask
# =================================================================================
def test_lnsp_to_sp_round_trip(engine):
    L = GOLDEN["lnsp_to_sp"]
    lnsp = np.array(L["lnsp"])
    source = test_source([{"param": "lnsp", "levelist": 1, "values": lnsp, **L["metadata"]}])
    sp_source = source | create_filter_by_name("lnsp_to_sp")
    sp = collect_fields_by_param(sp_source)
    assert set(sp) == {"sp"}
    assert np.allclose(sp["sp"][0].to_numpy(), np.exp(lnsp))
    assert sp["sp"][0].metadata("levelist") is None  # R: lnsp_to_sp.py:45
    back = collect_fields_by_param(sp_source | create_filter_by_name("sp_to_lnsp"))
    assert set(back) == {"lnsp"}
    assert np.allclose(back["lnsp"][0].to_numpy(), lnsp)


@pytest.mark.parametrize("name", ["impute_nans", "replace_nans", "impute_nans_fields"])
@pytest.mark.parametrize("params,value", [("t", 0.0), (["t", "q"], -1.0), ("r", 0.0)])
def test_impute_nans(engine, name, params, value):
    I = GOLDEN["impute_nans"]
    source = test_source([{"param": p, "values": nan_rows(I[p]), **I["metadata"]} for p in ("t", "q", "r")])
    inputs = collect_fields_by_param(source)
    outputs = collect_fields_by_param(source | create_filter_by_name(name, param=params, value=value))
    selected = [params] if isinstance(params, str) else params
    for p in ("t", "q", "r"):
        original = inputs[p][0].to_numpy(flatten=True)
        result = outputs[p][0].to_numpy(flatten=True)
        if p in selected:
            expected = original.copy()
            expected[np.isnan(expected)] = value
            assert np.array_equal(result, expected) and not np.any(np.isnan(result))
            assert np.array_equal(outputs[p][0].grid_points()[0], inputs[p][0].grid_points()[0])
        else:
            assert np.array_equal(result, original, equal_nan=True)


@pytest.mark.parametrize("name", ["clip", "clipper", "clip_fields"])
@pytest.mark.parametrize("bounds", [dict(minimum=2.5), dict(maximum=4.0), dict(minimum=2.5, maximum=4.0)])
def test_clip(engine, name, bounds):
    x = np.array([[1.0, np.nan, 3.0], [-0.0, 5.0, 6.0], [7.0, 2.5, 4.0]])
    md = GOLDEN["impute_nans"]["metadata"]
    source = test_source([{"param": "tp", "values": x, **md}, {"param": "q", "values": x, **md}])
    out = collect_fields_by_param(source | create_filter_by_name(name, param="tp", **bounds))
    assert np.array_equal(out["tp"][0].to_numpy(), np.clip(x, bounds.get("minimum"), bounds.get("maximum")), equal_nan=True)
    assert out["q"][0] is source.ds[1]


def test_clip_needs_a_bound():
    with pytest.raises(ValueError, match="At least one value for minimum or maximum"):
        create_filter_by_name("clip", param="tp")


def test_rescale_and_convert(engine):
    """R: tests/field_filters/test_rescale.py:17-72 (remote GRIB there; synthetic 2t/sp here)."""
    rng = np.random.default_rng(3)
    md = dict(latitudes=np.linspace(90, -90, 32), longitudes=np.arange(64) * 5.625, valid_datetime="2020-01-01T00:00:00Z")
    t2 = 280.0 + 10 * rng.standard_normal((32, 64))
    sp = 1e5 + 100 * rng.standard_normal((32, 64))
    source = test_source([{"param": "2t", "values": t2, **md}, {"param": "sp", "values": sp, **md}])

    rescale = create_filter_by_name("rescale", scale=1.0, offset=-273.15, param="2t")
    out = collect_fields_by_param(source | rescale)
    np.testing.assert_allclose(out["2t"][0].to_numpy(), t2 - 273.15)
    assert np.array_equal(out["2t"][0].to_numpy(), oracle.rescale_forward(t2, 1.0, -273.15))
    assert out["sp"][0] is source.ds[1]
    back = collect_fields_by_param((source | rescale) | rescale.reverse())
    assert np.array_equal(back["2t"][0].to_numpy(), oracle.rescale_backward(oracle.rescale_forward(t2, 1.0, -273.15), 1.0, -273.15))

    convert = create_filter_by_name("convert", unit_in="K", unit_out="degC", param="2t")
    out = collect_fields_by_param(source | convert)
    np.testing.assert_allclose(out["2t"][0].to_numpy(), t2 - 273.15)
    assert out["2t"][0].metadata("units") == "degC"
    back = collect_fields_by_param((source | convert) | convert.reverse())
    np.testing.assert_allclose(back["2t"][0].to_numpy(), t2)


def test_rescale_validation():
    with pytest.raises(TypeError, match="Missing required input"):
        create_filter_by_name("rescale", scale=1.0, param="2t")
    with pytest.raises(ValueError, match=r"Unknown input\(s\)"):
        create_filter_by_name("rescale", scale=1.0, offset=0.0, param="2t", extra=1)


def test_glacier_mask(engine, tmp_path):
    """R: tests/field_filters/test_glacier_mask.py."""
    md = GOLDEN["orog_to_z"]["metadata"]
    sd = np.array([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])
    gm = np.array([[0, 1], [0, 0], [1, 0]])
    path = tmp_path / "glacier.npy"
    np.save(path, gm)
    source = test_source([{"param": "sd", "values": sd, **md}, {"param": "2t", "values": sd, **md}])
    out = collect_fields_by_param(source | create_filter_by_name("glacier_mask", glacier_mask=str(path)))
    assert set(out) == {"sd_masked", "2t"}
    expected = sd.copy()
    expected[gm.astype(bool)] = np.nan
    assert np.array_equal(out["sd_masked"][0].to_numpy(), expected, equal_nan=True)
    assert out["sd_masked"][0].metadata("units") == "Fraction"


# =================================================================================
# regrid (R: tests/field_filters/test_regrid.py is smoke-only; numerics vs the oracle)
# =================================================================================
def synthetic_fields(grid, n, seed=0, nan_frac=0.0):
    rng = np.random.default_rng(seed)
    lat, lon = np.deg2rad(grid["latitudes"]), np.deg2rad(grid["longitudes"])
    specs = []
    for l in range(n):
        v = 280 + 30 * np.sin(lat) * np.cos(2 * lon + 0.1 * l) + rng.standard_normal(len(lat))
        if nan_frac:
            v[rng.random(len(v)) < nan_frac] = np.nan
        specs.append({"param": "t", "levelist": l + 1, "values": v, "latitudes": grid["latitudes"], "longitudes": grid["longitudes"],
                      "valid_datetime": "2020-01-01T00:00:00Z"})
    return specs


def test_regrid_nearest(engine):
    """R: regrid.py:315-381 — config C2-like: O32 -> 5 degree lat-lon, k = 1, bit-exact."""
    from anemoi_transform_amd.grids import lookup

    src, tgt = lookup("o32"), lookup([5.0, 5.0])
    specs = synthetic_fields(src, 5, nan_frac=0.02)
    source = test_source(specs)
    regrid = create_filter_by_name("regrid", in_grid="o32", out_grid=[5.0, 5.0], method="nearest")
    out = list(source | regrid)
    want = oracle.filter_regrid_nearest([dict(s) for s in specs], in_grid=src, out_grid=tgt)
    assert len(out) == 5
    for f, w in zip(out, want):
        assert np.array_equal(f.to_numpy(flatten=True), w["values"], equal_nan=True)
        lat, lon = f.grid_points()
        assert np.array_equal(lat, tgt["latitudes"]) and np.array_equal(lon, tgt["longitudes"])
        assert f.metadata("param") == "t"
    assert [f.metadata("levelist") for f in out] == [1, 2, 3, 4, 5]
    # in_grid defaults to the first field's own grid (R: regrid.py:359-361)
    out2 = list(source | create_filter_by_name("regrid", out_grid=tgt, method="nearest"))
    assert np.array_equal(out2[3].to_numpy(), out[3].to_numpy(), equal_nan=True)


def test_regrid_nearest_checks(engine):
    from anemoi_transform_amd.grids import lookup

    with pytest.raises(ValueError, match="out_grid is required"):
        create_filter_by_name("regrid", in_grid="o32", method="nearest")
    with pytest.raises(NotImplementedError, match="matrix"):  # only 'linear' has an in-tree stand-in for earthkit-regrid
        create_filter_by_name("regrid", in_grid="o32", out_grid="o48", method="conservative")
    source = test_source(synthetic_fields(lookup("o32"), 1))
    wrong = create_filter_by_name("regrid", in_grid="o48", out_grid=[5.0, 5.0], method="nearest")
    with pytest.raises(AssertionError):  # R: regrid.py:377-378
        list(source | wrong)


def test_regrid_wrong_field_size_raises_what_the_reference_raises(engine):
    """One exception type per interpolator, each the type of the reference's own statement on a field of the wrong length:
    ``matrix=`` -> ValueError (scipy's ``csr_array @ x``, R: regrid.py:310), ``method="nearest"`` -> AssertionError
    (R: regrid.py:377-378), ``mask=`` -> IndexError (numpy's ``x[..., mask]``, R: regrid.py:420)."""
    import scipy.sparse

    from anemoi_transform_amd import interp
    from anemoi_transform_amd.grids import lookup

    src, other, tgt = lookup("o32"), lookup("o48"), lookup([5.0, 5.0])
    source = test_source(synthetic_fields(src, 2))
    n_src, n_other = len(src["latitudes"]), len(other["latitudes"])

    # the reference's statements themselves, on the same sizes
    with pytest.raises(ValueError, match="dimension mismatch"):
        scipy.sparse.csr_array((np.ones(2), np.array([0, 1]), np.array([0, 1, 2])), shape=(2, n_other)) @ np.zeros(n_src)
    with pytest.raises(IndexError):
        np.zeros(n_src)[..., np.zeros(n_other, dtype=bool)]
    with pytest.raises(IndexError):
        np.zeros(n_src)[..., np.array([0, n_other - 1])]

    idx, w = interp.knn_inverse_distance(other, tgt, k=4)
    n_tgt = len(idx)
    matrix = dict(matrix_data=w.reshape(-1), matrix_indices=idx.reshape(-1).astype(np.int32), matrix_indptr=np.arange(n_tgt + 1, dtype=np.int32) * 4,
                  matrix_shape=np.array([n_tgt, n_other]))
    matrix = {**matrix, "in_latitudes": other["latitudes"], "in_longitudes": other["longitudes"],
              "out_latitudes": tgt["latitudes"], "out_longitudes": tgt["longitudes"]}
    with pytest.raises(ValueError, match="dimension mismatch"):
        list(source | create_filter_by_name("regrid", matrix=matrix))
    with pytest.raises(AssertionError):
        list(source | create_filter_by_name("regrid", in_grid="o48", out_grid=[5.0, 5.0], method="nearest"))
    with pytest.raises(IndexError):
        list(source | create_filter_by_name("regrid", mask=np.zeros(n_other, dtype=bool)))
    with pytest.raises(IndexError):
        list(source | create_filter_by_name("regrid", mask=np.array([0, n_other - 1])))
    with pytest.raises(IndexError, match="integer"):  # numpy: "arrays used as indices must be of integer (or boolean) type"
        np.zeros(n_src)[..., np.array([0.0, 1.0])]
    with pytest.raises(IndexError, match="integer"):
        list(source | create_filter_by_name("regrid", mask=np.array([0.0, 1.0])))
    # and the right sizes go through
    assert len(list(test_source(synthetic_fields(other, 2)) | create_filter_by_name("regrid", matrix=matrix))) == 2


def test_regrid_default_route(engine, tmp_path, caplog):
    """R: regrid.py:455-467 + 211-259 — no ``method`` (the reference's stock recipe form) resolves to the default
    interpolator with method="linear"; here the matrix is the in-tree bilinear one (earthkit-regrid's inventory is
    remote), applied like ``matrix=``: equal, bit for bit, to the oracle's csr statement on that same matrix."""
    import logging

    from anemoi_transform_amd import interp
    from anemoi_transform_amd.filters.regrid import EarthkitRegrid, _interpolator
    from anemoi_transform_amd.grids import lookup

    assert _interpolator() == "EarthkitRegrid" == _interpolator(method="linear")
    src, tgt = lookup("o32"), lookup([5.0, 5.0])
    specs = synthetic_fields(src, 4)
    with caplog.at_level(logging.WARNING):
        regrid = create_filter_by_name("regrid", in_grid="O32", out_grid=[5.0, 5.0])
    assert isinstance(regrid.interpolator, EarthkitRegrid) and regrid.interpolator.method == "linear"
    assert any("bilinear" in r.message and "MIR" in r.message for r in caplog.records)  # parity vs MIR is unpinned: said once
    out = list(test_source(specs) | regrid)
    matrix = interp.bilinear_octahedral(32, tgt)
    want = oracle.filter_regrid_matrix([dict(s) for s in specs], matrix={**matrix, "out_latitudes": tgt["latitudes"], "out_longitudes": tgt["longitudes"]})
    assert len(out) == 4
    for f, w_ in zip(out, want):
        assert np.array_equal(f.to_numpy(flatten=True), w_["values"])
        lat, lon = f.grid_points()
        assert np.array_equal(lat, tgt["latitudes"]) and np.array_equal(lon, tgt["longitudes"]) and f.metadata("param") == "t"
    # config form, other row-structured sources, explicit method
    for in_grid, name in (("f16", "f16"), ([10.0, 10.0], [10.0, 10.0])):
        g = lookup(name)
        lin = [{"param": "t", "values": g["latitudes"] * 2.0 + 1.0, "latitudes": g["latitudes"], "longitudes": g["longitudes"]}]
        got = list(test_source(lin) | create_filter(None, {"regrid": {"in_grid": in_grid, "out_grid": "o16", "method": "linear"}}))[0]
        o16 = lookup("o16")
        inside = np.abs(o16["latitudes"]) < 80
        np.testing.assert_allclose(got.to_numpy()[inside], o16["latitudes"][inside] * 2.0 + 1.0, rtol=1e-12)  # linear in latitude: reproduced
    # what the in-tree default cannot serve fails loudly, at construction
    path = str(tmp_path / "grid.npz")
    np.savez(path, latitudes=src["latitudes"], longitudes=src["longitudes"])
    with pytest.raises(NotImplementedError, match="matrix"):
        create_filter_by_name("regrid", in_grid=path, out_grid=[5.0, 5.0])
    with pytest.raises(TypeError):  # in_grid is a required argument of the default interpolator, as in the reference
        create_filter_by_name("regrid", out_grid=[5.0, 5.0])
    with pytest.raises(ValueError, match="points"):
        list(test_source(synthetic_fields(lookup("o16"), 1)) | regrid)


@pytest.mark.parametrize("kind", ["bilinear_k4", "knn_k3", "ragged_csr"])
def test_regrid_matrix(engine, tmp_path, kind):
    """R: regrid.py:262-312 — npz matrix in the make-regrid-file format against scipy's csr_matvec (f64: bit-exact)."""
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.grids import lookup

    src, tgt = lookup("o32"), lookup([5.0, 5.0])
    if kind == "bilinear_k4":
        matrix = interp.bilinear_octahedral(32, tgt)
    elif kind == "knn_k3":
        idx, w = interp.knn_inverse_distance(src, tgt, k=3)
        matrix = interp.ell_to_csr(idx, w, len(src["latitudes"]))
    else:
        idx, w = interp.knn_inverse_distance(src, tgt, k=4)
        keep = (np.arange(idx.size) % 5 != 0).reshape(idx.shape)  # drop entries -> rows of 3 or 4
        lengths = keep.sum(axis=1)
        matrix = dict(matrix_data=w[keep], matrix_indices=idx[keep].astype(np.int32),
                      matrix_indptr=np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32),
                      matrix_shape=np.array([len(idx), len(src["latitudes"])]))
    path = str(tmp_path / "matrix.npz")
    interp.save_matrix_npz(path, matrix, src, tgt)
    specs = synthetic_fields(src, 4)
    out = list(test_source(specs) | create_filter(None, {"regrid": {"matrix": path}}))
    want = oracle.filter_regrid_matrix([dict(s) for s in specs], matrix=dict(np.load(path)))
    for f, w_ in zip(out, want):
        assert np.array_equal(f.to_numpy(flatten=True), w_["values"])
        assert np.array_equal(f.grid_points()[0], tgt["latitudes"])
    if kind == "bilinear_k4":
        # bilinear weights reproduce a field that is linear in latitude
        lin = [{"param": "t", "values": src["latitudes"] * 2.0 + 1.0, "latitudes": src["latitudes"], "longitudes": src["longitudes"]}]
        got = list(test_source(lin) | create_filter_by_name("regrid", matrix=path))[0].to_numpy()
        inside = np.abs(tgt["latitudes"]) < 87
        np.testing.assert_allclose(got[inside], tgt["latitudes"][inside] * 2.0 + 1.0, rtol=1e-12)


@pytest.mark.parametrize("as_bool", [False, True])
def test_regrid_mask(engine, tmp_path, as_bool):
    """R: regrid.py:384-429 — index / boolean subset, lat/lon from the first field."""
    from anemoi_transform_amd.grids import lookup

    src = lookup("o32")
    n = len(src["latitudes"])
    rng = np.random.default_rng(9)
    index = np.sort(rng.choice(n, size=n // 3, replace=False))
    mask = np.zeros(n, bool)
    mask[index] = True
    path = str(tmp_path / "mask.npz")
    np.savez(path, mask=mask if as_bool else index)
    specs = synthetic_fields(src, 3, nan_frac=0.05)
    out = list(test_source(specs) | create_filter_by_name("regrid", mask=path))
    want = oracle.filter_regrid_mask([dict(s) for s in specs], mask=mask if as_bool else index)
    for f, w in zip(out, want):
        assert np.array_equal(f.to_numpy(flatten=True), w["values"], equal_nan=True)
        assert np.array_equal(f.grid_points()[0], w["latitudes"]) and np.array_equal(f.grid_points()[1], w["longitudes"])


def test_regrid_shards_concatenate_to_the_full_result(engine):
    """Target-point sharding: the rank slices are disjoint rows of the same operator."""
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.grids import lookup

    src, tgt = lookup("o32"), lookup([5.0, 5.0])
    idx, w = interp.knn_inverse_distance(src, tgt, k=4)
    matrix = {**interp.ell_to_csr(idx, w, len(src["latitudes"])), "out_latitudes": tgt["latitudes"], "out_longitudes": tgt["longitudes"]}
    source = test_source(synthetic_fields(src, 3))
    full = list(source | create_filter_by_name("regrid", matrix=matrix))
    parts = [list(source | create_filter_by_name("regrid", matrix=matrix, shard=(r, 3))) for r in range(3)]
    for l in range(3):
        assert np.array_equal(np.concatenate([p[l].to_numpy() for p in parts]), full[l].to_numpy())
        assert np.array_equal(np.concatenate([p[l].grid_points()[0] for p in parts]), tgt["latitudes"])


def test_chained_filters_config5_shape(engine):
    """regrid -> orog_to_z -> convert (config 5 of BASELINE.json) against the chained oracle."""
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.grids import lookup

    src, tgt = lookup("o32"), lookup([5.0, 5.0])
    idx, w = interp.knn_inverse_distance(src, tgt, k=4)
    matrix = {**interp.ell_to_csr(idx, w, len(src["latitudes"])), "out_latitudes": tgt["latitudes"], "out_longitudes": tgt["longitudes"]}
    specs = synthetic_fields(src, 3)
    specs[1]["param"] = "orog"
    specs[2]["param"] = "q"
    pipeline = (
        test_source(specs)
        | create_filter_by_name("regrid", matrix=matrix)
        | create_filter_by_name("orog_to_z")
        | create_filter_by_name("convert", unit_in="K", unit_out="degC", param="t")
    )
    out = collect_fields_by_param(pipeline)
    assert set(out) == {"t", "z", "q"}
    base = oracle.filter_regrid_matrix([dict(s) for s in specs], matrix=matrix)
    assert np.array_equal(out["q"][0].to_numpy(), base[2]["values"])
    assert np.array_equal(out["z"][0].to_numpy(), oracle.orog_to_z(base[1]["values"]))
    assert np.array_equal(out["t"][0].to_numpy(), oracle.rescale_forward(base[0]["values"], 1.0, -273.15))


def test_config1_32x64_single_field_plumbing(engine):
    """BASELINE.json configs[0]: one 32x64 lat-lon field through registered field filters."""
    rng = np.random.default_rng(20260630)
    lat = np.linspace(90, -90, 32)
    lon = np.arange(64) * 5.625
    t2 = 280 + 5 * rng.standard_normal((32, 64))
    lsm = (np.sin(3 * np.deg2rad(lat))[:, None] * np.cos(2 * np.deg2rad(lon))[None, :] > 0).astype(float)
    specs = [{"param": "2t", "values": t2, "latitudes": lat, "longitudes": lon, "valid_datetime": "2020-01-01T00:00:00Z"},
             {"param": "lsm", "values": lsm, "latitudes": lat, "longitudes": lon, "valid_datetime": "2020-01-01T00:00:00Z"}]
    pipeline = test_source(specs) | create_filter_by_name("rescale", scale=1, offset=-273.15, param="2t") | create_filter_by_name(
        "apply_mask", mask_param="lsm", mask_value=0
    )
    out = list(pipeline)
    assert len(out) == 1 and out[0].metadata("param") == "2t" and out[0].shape == (32, 64)
    want = oracle.filter_apply_mask(oracle.filter_rescale([dict(s) for s in specs], scale=1, offset=-273.15, param="2t"),
                                    mask_param="lsm", mask_value=0)
    assert np.array_equal(out[0].to_numpy(flatten=True), want[0]["values"], equal_nan=True)


def test_every_registered_filter_is_a_factory():
    """R: tests/test_create.py:15-23."""
    expected = {"regrid", "apply_mask_fields", "mask", "remove_nans_fields", "remove_nans", "rescale", "convert", "orog_to_z_fields",
                "z_to_orog_fields", "geopotential_to_height", "height_to_geopotential", "clip_fields", "clip", "impute_nans_fields",
                "impute_nans", "lnsp_to_sp", "sp_to_lnsp", "glacier_mask", "noop"}
    assert expected <= set(filter_registry.registered)
    for alias, target in {"apply_mask": "mask", "drop_nans": "remove_nans", "orog_to_z": "geopotential_to_height",
                          "z_to_orog": "height_to_geopotential", "clipper": "clip", "replace_nans": "impute_nans"}.items():
        assert filter_registry.lookup(alias) is filter_registry.lookup(target)
    assert create_filter_by_name("noop", context="ctx").context == "ctx"


class ForeignField:
    """A field that is NOT one of ours — the surface an earthkit-data field offers to the filters
    (SURVEY.md §8b): to_numpy / metadata / grid_points / shape."""

    def __init__(self, values, lat, lon, **md):
        self._v, self._lat, self._lon, self._md = np.asarray(values), lat, lon, md
        self.shape = self._v.shape

    def to_numpy(self, flatten=False, dtype=None, index=None):
        v = self._v.astype(dtype) if dtype is not None else self._v.copy()
        return v.flatten() if flatten else v

    def grid_points(self):
        return self._lat, self._lon

    def metadata(self, *keys, namespace=None, default=None, **kw):
        if namespace:
            return {}
        if not keys:
            md = self._md

            class View:
                def get(self, key, default=None):
                    return md.get(key, default)

                def keys(self):
                    return md.keys()

                def __getitem__(self, key):
                    return md[key]

            return View()
        out = []
        for k in keys:
            if k not in self._md:
                raise KeyError(k)
            out.append(self._md[k])
        return out[0] if len(out) == 1 else tuple(out)


def test_foreign_fields_are_accepted(engine):
    """Fields from another library (earthkit-data when installed) go through the same filters."""
    from anemoi_transform_amd.fields import FieldList
    from anemoi_transform_amd.grids import lookup

    src, tgt = lookup("o16"), lookup([20.0, 20.0])
    rng = np.random.default_rng(2)
    fields = FieldList([ForeignField(280 + rng.standard_normal(len(src["latitudes"])), src["latitudes"], src["longitudes"],
                                     param=p, levelist=l) for p, l in (("t", 500), ("orog", 0), ("t", 850))])
    out = (create_filter_by_name("regrid", out_grid=tgt, method="nearest") | create_filter_by_name("orog_to_z")
           | create_filter_by_name("rescale", scale=1.0, offset=-273.15, param="t")).forward(fields)
    idx = oracle.nearest_grid_points(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"])
    assert [f.metadata("param") for f in out] == ["t", "z", "t"] and [f.metadata("levelist") for f in out] == [500, 0, 850]
    assert np.array_equal(out[0].to_numpy(), oracle.rescale_forward(fields[0].to_numpy()[idx], 1.0, -273.15))
    assert np.array_equal(out[1].to_numpy(), oracle.orog_to_z(fields[1].to_numpy()[idx]))
    assert np.array_equal(out[2].grid_points()[0], tgt["latitudes"])
    assert out[0].metadata().get("param") == "t" and "levelist" in list(out[0].metadata().keys())


def test_long_field_lists_are_split_into_bounded_stacks(engine, monkeypatch):
    """Hundreds of host fields on one grid (BASELINE config 4 has 3288) become several stacks of bounded size; results and
    order are those of the field-by-field loop."""
    from anemoi_transform_amd import fields as fields_module
    from anemoi_transform_amd.grids import lookup

    monkeypatch.setattr(fields_module, "MAX_STACK_LEVELS", 7)
    src, tgt = lookup("o16"), lookup([10.0, 10.0])
    specs = synthetic_fields(src, 17)
    for i, s_ in enumerate(specs):
        s_["param"] = "t" if i % 3 else "q"
    groups = fields_module.group_into_stacks(list(test_source(specs)))
    positions = [p for g in groups for p in g.positions]
    assert [g.stack.n_lev for g in groups] == [7, 7, 3] and sorted(positions) == list(range(17))
    # inside the stacks the fields of one variable sit next to each other (a list that alternates between variables would give a
    # per-variable filter a program that changes at every level; this way it has one run per variable) — which level a field becomes
    # is internal: the results below come back in the order of the list
    assert [specs[p]["param"] for p in positions] == ["q"] * 6 + ["t"] * 11
    assert positions == [p for p in range(17) if p % 3 == 0] + [p for p in range(17) if p % 3]
    regrid = create_filter_by_name("regrid", in_grid="o16", out_grid=[10.0, 10.0], method="nearest")
    rescale = create_filter_by_name("rescale", scale=2.0, offset=1.0, param="t")
    out = list(test_source(specs) | regrid | rescale)
    want = oracle.filter_rescale(oracle.filter_regrid_nearest([dict(s_) for s_ in specs], in_grid=src, out_grid=tgt), scale=2.0, offset=1.0, param="t")
    assert len(out) == 17
    for f, w, spec in zip(out, want, specs):
        assert f.metadata("levelist") == spec["levelist"] and f.metadata("param") == spec["param"]
        assert np.array_equal(f.to_numpy(flatten=True), np.asarray(w["values"]).ravel(), equal_nan=True)
