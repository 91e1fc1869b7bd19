"""Metadata / re-listing filters (rename, clear_step, repeat_members, earthkitfieldlambda, empty,
icon_refinement_level), written like the reference's tests:
R: tests/field_filters/test_rename.py, test_clear_step.py, test_repeat_members.py, test_lambda.py.
What this package adds on top: the data of a device field stays where it is.
"""

from __future__ import annotations

import datetime

import numpy as np
import pytest

from anemoi_transform_amd import interp, native
from anemoi_transform_amd.fields import FieldList, to_datetime
from anemoi_transform_amd.filters import create_filter_by_name as create_filter
from anemoi_transform_amd.grids import lookup

import native_double
from test_filters import collect_fields_by_param, synthetic_fields, test_source


@pytest.fixture(params=["double", pytest.param("hip", marks=pytest.mark.gpu)])
def engine(request, monkeypatch):
    if request.param == "double":
        native_double.install(monkeypatch)
    return request.param


LATLON = {"latitudes": [10.0, 0.0, -10.0], "longitudes": [20, 40.0], "valid_datetime": "2018-08-01T12:00:00Z"}
VALUES = np.array([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])


def pl_source():
    specs = []
    for param in ("z", "t"):
        for level in (1000, 850, 500):
            specs.append({"param": param, "levelist": level, "levtype": "pl", "values": VALUES + level, **LATLON})
    return test_source(specs)


def test_rename_by_table():
    """R: tests/field_filters/test_rename.py:28-42, 60-77."""
    source = pl_source()
    rename = create_filter("rename", param={"z": "geopotential", "t": "temperature"},
                           levelist={1000: "1000hPa", 850: "850hPa", 500: "500hPa"})
    for original, result in zip(source, source | rename):
        assert result.metadata("levelist") == f"{original.metadata('levelist')}hPa"
        assert result.metadata("param") == {"z": "geopotential", "t": "temperature"}[original.metadata("param")]
        assert np.array_equal(result.to_numpy(), original.to_numpy())
    # a value outside the table, or a key the field does not have, leaves the field itself in place
    other = create_filter("rename_fields", param={"q": "humidity"}, number={1: 2})
    for original, result in zip(source, source | other):
        assert result is original


def test_rename_by_format():
    """R: tests/field_filters/test_rename.py:45-57 — typed access ``{levelist:d}`` included."""
    source = pl_source()
    rename = create_filter("rename", param="{param}_{levelist}_{levtype}_{levelist:d}")
    for original, result in zip(source, source | rename):
        p, lev, levtype, lev_d = original.metadata("param", "levelist", "levtype", "levelist:d")
        assert isinstance(lev, int) and isinstance(lev_d, float)
        assert result.metadata("param") == f"{p}_{lev}_{levtype}_{lev_d}"
    with pytest.raises(ValueError, match="Invalid value for rename"):
        create_filter("rename", param=3)
    with pytest.raises(NotImplementedError):
        create_filter("rename", columns={"a": "b"})


def test_rename_sees_earlier_renames():
    source = pl_source()
    rename = create_filter("rename", param={"t": "temperature"}, levelist="{param}@{levelist}")
    out = collect_fields_by_param(source | rename)
    assert sorted(f.metadata("levelist") for f in out["temperature"]) == ["temperature@1000", "temperature@500", "temperature@850"]


def test_clear_step():
    """R: tests/field_filters/test_clear_step.py:37-60."""
    source = test_source([{"param": "t", "step": s, "values": VALUES, **LATLON} for s in (0, 6, 12)])
    for original, result in zip(source, source | create_filter("clear_step")):
        expected = to_datetime(original.metadata("valid_datetime")) - datetime.timedelta(hours=original.metadata("step"))
        assert to_datetime(result.metadata("valid_datetime")) == expected
        assert result.metadata("step") == 0
        assert result.metadata("date") == int(expected.strftime("%Y%m%d")) and result.metadata("time") == int(expected.strftime("%H%M"))
        assert np.array_equal(original.to_numpy(), result.to_numpy())


@pytest.mark.parametrize("config,numbers", [
    (dict(numbers=[1, 2, 3]), [1, 2, 3]), (dict(numbers="1/to/3"), [1, 2, 3]), (dict(members=[0, 2, 4]), [1, 3, 5]),
    (dict(members="0/to/2"), [1, 2, 3]), (dict(count=4), [1, 2, 3, 4]), (dict(numbers="1/to/5/by/2"), [1, 3, 5]),
])
def test_repeat_members(config, numbers):
    """R: tests/field_filters/test_repeat_members.py — every field once per member, values shared, ``number`` set."""
    source = test_source([{"param": "2t", "name": "2 metre temperature", "values": VALUES, **LATLON},
                          {"param": "msl", "name": "pressure", "values": VALUES * 2, **LATLON}])
    repeated = create_filter("repeat_members", **config).forward(list(source))
    assert len(repeated) == 2 * len(numbers)
    for i, f in enumerate(repeated):
        original = list(source)[i // len(numbers)]
        assert f.metadata("number") == numbers[i % len(numbers)]
        assert f.metadata("name") == original.metadata("name")
        assert np.array_equal(f.values, original.values)


def test_repeat_members_validation():
    for bad in (dict(), dict(numbers=[1], count=2), dict(members=[0], numbers=[1])):
        with pytest.raises(ValueError, match="Exactly one of members, count or numbers"):
            create_filter("repeat_members", **bad)


def times_a(field, a):
    return field.clone(values=field.values * a)


def divided_by_a(field, a):
    return field.clone(values=field.values / a)


def test_earthkitfieldlambda():
    """R: tests/field_filters/test_lambda.py:56-90."""
    source = test_source([{"param": "sp", "values": VALUES * 1000, **LATLON}, {"param": "2t", "values": VALUES + 270, **LATLON}])
    fieldlist = list(source)
    f = create_filter("earthkitfieldlambda", fn="test_metadata_filters.times_a", param="sp", fn_args=[10],
                      backward_fn="test_metadata_filters.divided_by_a")
    forward = f.forward(fieldlist)
    back = f.backward(forward)
    for before, mid, after in zip(fieldlist, forward, back):
        np.testing.assert_allclose(after.to_numpy(), before.to_numpy())
        if before.metadata("param") == "sp":
            np.testing.assert_allclose(mid.to_numpy(), before.to_numpy() * 10)
            assert mid.to_numpy().shape == VALUES.shape
        else:
            assert mid is before
    with pytest.raises(ValueError, match="Could not import function"):
        create_filter("earthkitfieldlambda", fn="no.such.function", param="sp")
    with pytest.raises(ValueError, match="Expected 'fn_args' to be a list"):
        create_filter("earthkitfieldlambda", fn="test_metadata_filters.times_a", param="sp", fn_args=3)
    with pytest.raises(TypeError, match="Missing required input"):
        create_filter("earthkitfieldlambda", fn="test_metadata_filters.times_a")
    one_way = create_filter("earthkitfieldlambda", fn="test_metadata_filters.times_a", param="sp", fn_args=[2])
    with pytest.raises(ValueError, match="Backward function is undefined"):
        one_way.backward(fieldlist)


def test_empty():
    assert len(create_filter("empty").forward(list(pl_source()))) == 0


def test_icon_refinement_level(engine, tmp_path):
    """R: icon_refinement_level.py:56-85 — k = 1 gather to the ICON cells up to a refinement level."""
    src = lookup("o32")
    rng = np.random.default_rng(5)
    n_cells = 500
    clat, clon = np.arcsin(rng.uniform(-1, 1, n_cells)), rng.uniform(-np.pi, np.pi, n_cells)
    level = rng.integers(0, 4, n_cells)
    path = str(tmp_path / "icon_grid.npz")
    np.savez(path, clat=clat, clon=clon, refinement_level_c=level)
    specs = synthetic_fields(src, 4, nan_frac=0.01)
    out = list(test_source(specs) | create_filter("icon_refinement_level", grid=path, refinement_level_c=2))
    keep = level <= 2
    lat, lon = np.rad2deg(clat[keep]), np.rad2deg(clon[keep])
    nearest = interp.nearest_grid_points(src["latitudes"], src["longitudes"], lat, lon)
    assert len(out) == 4
    for spec, f in zip(specs, out):
        assert np.array_equal(f.to_numpy(flatten=True), spec["values"][nearest], equal_nan=True)
        assert np.array_equal(f.grid_points()[0], lat) and np.array_equal(f.grid_points()[1], lon)
        assert f.resolution == "mrl2" and f.metadata("levelist") == spec["levelist"]
    everything = list(test_source(specs) | create_filter("icon_refinement_level", grid=path, refinement_level_c=None))
    assert everything[0].to_numpy().shape == (n_cells,)


def test_rename_keeps_device_fields_on_the_device_and_fuses(engine, monkeypatch):
    """regrid | rename | rescale: one gather launch, the rename rides along as metadata, nothing is copied to the host."""
    src, tgt = lookup("o32"), lookup([5.0, 5.0])
    idx, w = interp.knn_inverse_distance(src, tgt, k=4)
    matrix = {**interp.ell_to_csr(idx, w, len(src["latitudes"])), "out_latitudes": tgt["latitudes"], "out_longitudes": tgt["longitudes"]}
    specs = synthetic_fields(src, 5)
    specs[3]["param"] = "q"
    counts = {"regrid_ell": 0, "pointwise_stack": 0}
    for name in counts:
        real = getattr(native, name)

        def wrapped(*a, _real=real, _name=name, **k):
            counts[_name] += 1
            return _real(*a, **k)

        monkeypatch.setattr(native, name, wrapped)
    regrid = create_filter("regrid", matrix=matrix)
    rename = create_filter("rename", param={"t": "temperature"})
    rescale = create_filter("rescale", scale=1.0, offset=-273.15, param="temperature")
    fused = list(test_source(specs) | (regrid | rename | rescale))
    assert counts == {"regrid_ell": 1, "pointwise_stack": 0}
    monkeypatch.setenv("ATX_NO_FUSION", "1")
    plain = list(test_source(specs) | (regrid | rename | rescale))
    assert [f.metadata("param") for f in fused] == ["temperature"] * 3 + ["q", "temperature"] == [f.metadata("param") for f in plain]
    for a, b in zip(fused, plain):
        assert a.stack_ref() is not None and b.stack_ref() is not None
        assert np.array_equal(a.to_numpy(), b.to_numpy())
    # rename alone on device fields: same stack, same level, no launch
    counts.update(regrid_ell=0, pointwise_stack=0)
    renamed = rename.forward(FieldList(plain))
    assert counts == {"regrid_ell": 0, "pointwise_stack": 0}
    for a, b in zip(renamed, plain):
        assert a.stack_ref()[0] is b.stack_ref()[0] and a.stack_ref()[1] == b.stack_ref()[1]
