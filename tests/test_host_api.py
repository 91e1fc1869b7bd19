"""Host-side contract tests (no GPU): grids, registry, Transform / Filter bases, fields, C ABI exports.

Modelled on the offline parts of R: tests/test_filter.py, test_dispatchingfilter.py,
test_fields.py, test_create.py, test_grids.py.
"""

from __future__ import annotations

import ctypes
import json
import os
import re

import numpy as np
import pytest

from anemoi_transform_amd import grids, interp, native
from anemoi_transform_amd.core import (
    DispatchingFilter,
    Pipeline,
    Registry,
    ReversedTransform,
    SingleFieldFilter,
    Transform,
)
from anemoi_transform_amd.fields import (
    ArrayField,
    FieldList,
    FieldSelection,
    fieldlist_from_dicts,
    new_field_from_latitudes_longitudes,
    new_field_from_numpy,
)
from anemoi_transform_amd.gather import TARGET_COST, TARGET_COST_SHORT_LAUNCH, GatherPlan, equal_count_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))


# ---- grids -------------------------------------------------------------------------------
def test_lookup_o96_known_answers():
    """R: tests/test_grids.py:50-58."""
    g = GOLDEN["grid_o96"]
    x = grids.lookup("o96")
    assert x["latitudes"].shape == (g["n_points"],) == x["longitudes"].shape
    assert x["latitudes"].mean() == pytest.approx(g["mean_latitude"], abs=1e-9)
    assert x["longitudes"].mean() == pytest.approx(g["mean_longitude"])
    assert x["latitudes"][g["index"]] == pytest.approx(g["latitude_at_index"])
    assert x["longitudes"][g["index"]] == pytest.approx(g["longitude_at_index"])


@pytest.mark.parametrize("name,n", [("o32", 5248), ("O96", 40320), ("o1280", 6599680), ("f48", 96 * 192)])
def test_gaussian_grid_sizes(name, n):
    g = grids.lookup(name)
    assert len(g["latitudes"]) == n == len(g["longitudes"])
    assert np.all(np.diff(g["latitudes"]) <= 0)  # north to south
    assert g["longitudes"].min() == 0.0 and g["longitudes"].max() < 360.0


def test_latlon_grids_and_npz(tmp_path):
    g = grids.lookup([0.25, 0.25])
    assert len(g["latitudes"]) == 721 * 1440
    assert g["latitudes"][0] == 90.0 and g["latitudes"][-1] == -90.0 and g["longitudes"][1] == 0.25
    assert len(grids.lookup("1.0")["latitudes"]) == 181 * 360
    assert len(grids.lookup("5/2.5")["latitudes"]) == 37 * 144
    path = str(tmp_path / "grid.npz")
    np.savez(path, latitudes=g["latitudes"][:10], longitudes=g["longitudes"][:10])
    assert np.array_equal(grids.lookup(path)["longitudes"], g["longitudes"][:10])
    with pytest.raises(ValueError):
        grids.lookup("n320")  # classic reduced grids need the downloaded table (out of scope)


def test_n320_sized_reduced_grid():
    """BASELINE config 4's target: the SIZE of the classic N320 grid (542 080 points, SURVEY.md §8 header) on a locally
    constructed row table — even rows proportional to cos(lat), 18 at the poles, 1280 at the equator, symmetric."""
    rows = grids.sized_row_lengths(320, 542080)
    assert rows.shape == (640,) and rows.sum() == 542080
    assert np.array_equal(rows, rows[::-1]) and rows[0] == 18 and rows.max() == 1280 == rows[319]
    assert np.all(np.diff(rows[:320]) >= 0) and np.all(rows % 2 == 0)
    pinned = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "n320_sized_rows.json")))
    assert rows[:320].tolist() == pinned["north_rows"] and pinned["total"] == 542080  # the construction is pinned
    g = grids.lookup("n320-sized")
    assert len(g["latitudes"]) == 542080 == len(g["longitudes"])
    assert np.array_equal(np.unique(g["latitudes"])[::-1], grids.gaussian_latitudes(640))
    assert g["longitudes"][18] == 0.0 and g["longitudes"][1] == 20.0 and g["longitudes"].max() < 360.0
    # any admissible total is hit exactly, leftovers included
    for total in (2 * 48 * 40 + 6, 2 * 48 * 18, 8 * 48 * 48):
        assert grids.sized_row_lengths(48, total).sum() == total
    with pytest.raises(ValueError):
        grids.sized_row_lengths(48, 10)
    with pytest.raises(ValueError):
        grids.reduced_gaussian_sized(640)  # size of the classic N640 grid is not recorded


def test_gaussian_latitudes_are_legendre_roots():
    lats = grids.gaussian_latitudes(64)
    x = np.sin(np.deg2rad(lats))
    p = np.polynomial.legendre.Legendre.basis(64)(x)
    assert np.max(np.abs(p)) < 1e-12


# ---- precompute -------------------------------------------------------------------------
def test_bilinear_weights_rows_sum_to_one_and_bracket():
    tgt = grids.lookup([10.0, 10.0])
    m = interp.bilinear_octahedral(16, tgt)
    w = m["matrix_data"].reshape(-1, 4)
    assert np.allclose(w.sum(axis=1), 1.0) and w.min() >= 0.0
    assert interp.csr_uniform_k(m["matrix_indptr"]) == 4
    src = grids.lookup("o16")
    idx = m["matrix_indices"].reshape(-1, 4)
    # the four corners lie on at most two latitude rows around the target latitude
    for t in (0, 100, len(tgt["latitudes"]) - 1):
        rows = np.unique(src["latitudes"][idx[t]])
        assert len(rows) <= 2
        if len(rows) == 2:
            assert rows.min() <= tgt["latitudes"][t] <= rows.max()


def test_knn_weights_and_matrix_file(tmp_path):
    src, tgt = grids.lookup("o16"), grids.lookup([20.0, 20.0])
    idx, w = interp.knn_inverse_distance(src, tgt, k=4)
    assert idx.shape == w.shape == (len(tgt["latitudes"]), 4)
    assert np.allclose(w.sum(axis=1), 1.0)
    m = interp.ell_to_csr(idx, w, len(src["latitudes"]))
    path = str(tmp_path / "m.npz")
    interp.save_matrix_npz(path, m, src, tgt)
    loaded = interp.load_matrix_npz(path)
    # the npz keys the reference reads (R: filters/fields/regrid.py:281-290)
    assert set(loaded) == {"matrix_data", "matrix_indices", "matrix_indptr", "matrix_shape", "in_latitudes", "in_longitudes",
                           "out_latitudes", "out_longitudes"}
    assert loaded["matrix_indices"].dtype == np.int32 and loaded["matrix_indptr"].dtype == np.int32
    assert tuple(loaded["matrix_shape"]) == (len(tgt["latitudes"]), len(src["latitudes"]))


def test_gather_plan_validation_and_sharding():
    with pytest.raises(ValueError, match="outside"):
        GatherPlan(10, 3, index=np.array([0, 10, 2]))  # cKDTree's "no neighbour" marker is rejected
    with pytest.raises(ValueError):
        GatherPlan(10, 3, index=np.array([[0, 1], [2, 3], [4, 5]]))  # k = 2 without weights
    plan = GatherPlan(10, 7, index=np.arange(14).reshape(7, 2) % 10, weights=np.ones((7, 2)) / 2)
    parts = [plan.shard(r, 3) for r in range(3)]
    assert sum(p.n_tgt for p in parts) == 7
    assert np.array_equal(np.concatenate([p.index for p in parts]), plan.index)
    b = plan.bounds(3)
    assert b[0] == 0 and b[-1] == 7 and all(b[i] <= b[i + 1] for i in range(3))
    csr = GatherPlan(10, 3, csr=(np.ones(5), np.array([1, 2, 3, 4, 5]), np.array([0, 2, 2, 5])))
    s = csr.shard(1, 2)
    assert s.n_tgt == csr.bounds(2)[2] - csr.bounds(2)[1] and s.indptr[0] == 0 and s.indptr[-1] == len(s.indices)
    both = [csr.shard(r, 2) for r in range(2)]
    assert np.array_equal(np.concatenate([p.indices for p in both]), csr.indices)
    assert [equal_count_bounds(10, r, 4) for r in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]
    m = GatherPlan.from_mask(np.array([True, False, True, True]))
    assert m.k == 1 and np.array_equal(m.index[:, 0], [0, 2, 3])


# ---- registry ------------------------------------------------------------------------------
def test_registry_surface():
    reg = Registry("pkg")

    @reg.register("alpha")
    class Alpha:
        def __init__(self, x=1):
            self.x = x

    reg.register("beta", lambda **kw: ("beta", kw), aliases=["b", "be-ta"])
    assert reg.registered == ["alpha", "beta"]
    assert reg.aliases() == {"beta": ["b", "be_ta"]}
    assert set(reg.factories) == {"alpha", "beta"}
    assert reg.create("alpha", x=3).x == 3
    assert reg.create("b", y=1) == ("beta", {"y": 1})
    assert reg.from_config("alpha").x == 1
    assert reg.from_config({"alpha": {"x": 5}}).x == 5
    assert reg.from_config({"be-ta": None}) == ("beta", {})
    assert reg.lookup("nope", return_none=True) is None
    with pytest.raises(ValueError):
        reg.create("nope")
    with pytest.raises(AssertionError):  # R: filters/__init__.py:30-33 relies on this
        reg.register("alpha", Alpha)
    with pytest.raises(AssertionError):
        reg.register("gamma", Alpha, aliases=["b"])
    with pytest.raises(ValueError):
        reg.from_config({"a": 1, "b": 2})


# ---- Transform / SingleFieldFilter / DispatchingFilter contracts ----------------------------------
def test_singlefieldfilter_contract():
    """R: tests/test_filter.py:22-105 (the offline part)."""
    with pytest.raises(TypeError, match="abstract method"):
        SingleFieldFilter()

    class BadRequired(SingleFieldFilter):
        required_inputs = "string_not_allowed"

        def forward_transform(self, field):
            pass

    with pytest.raises(TypeError, match="Required inputs must be a list or tuple"):
        BadRequired()

    class Foo(SingleFieldFilter):
        required_inputs = ("foo",)

        def forward_transform(self, field):
            pass

    with pytest.raises(ValueError, match=r"Unknown input\(s\)"):
        Foo(foo="bar", baz="qux")
    with pytest.raises(TypeError, match="Missing required input"):
        Foo()

    class Opt(SingleFieldFilter):
        optional_inputs = {"temperature": "2t"}

        def forward_transform(self, field):
            pass

    assert Opt().temperature == "2t"
    assert Opt(temperature="temperature").temperature == "temperature"
    with pytest.raises(AttributeError):
        Opt().missing

    class Prep(SingleFieldFilter):
        required_inputs = ("positive_number",)

        def prepare_filter(self):
            if self.positive_number < 0:
                raise ValueError("positive_number must be positive")

        def forward_transform(self, field):
            pass

    with pytest.raises(ValueError, match="positive_number must be positive"):
        Prep(positive_number=-1)


SPECS = [
    {"param": "t", "levelist": 500, "values": np.arange(6.0).reshape(3, 2), "latitudes": [10.0, 0.0, -10.0], "longitudes": [20.0, 40.0]},
    {"param": "q", "levelist": 850, "values": np.arange(6.0).reshape(3, 2) * 2, "latitudes": [10.0, 0.0, -10.0], "longitudes": [20.0, 40.0]},
]


def test_user_defined_singlefieldfilter_runs_on_host_arrays():
    """A filter written against the reference API (numpy in forward_transform) works unchanged
    (R: tests/test_filter.py:108-170)."""

    class PlusOne(SingleFieldFilter):
        optional_inputs = {"param": "t"}

        def forward_select(self):
            return {"param": self.param}

        def forward_transform(self, field):
            return self.new_field_from_numpy(field.to_numpy() + 1, template=field)

        def backward_transform(self, field):
            return self.new_field_from_numpy(field.to_numpy() - 1, template=field)

    fl = fieldlist_from_dicts(SPECS)
    out = PlusOne()(fl)
    assert np.array_equal(out[0].to_numpy(), SPECS[0]["values"] + 1)
    assert out[1] is fl[1]
    back = PlusOne().reverse()(out)
    assert np.array_equal(back[0].to_numpy(), SPECS[0]["values"])
    rev = PlusOne.reversed(param="q")
    assert isinstance(rev, ReversedTransform)
    assert np.array_equal(rev(fl)[1].to_numpy(), SPECS[1]["values"] - 1)

    class ForwardOnly(SingleFieldFilter):
        def forward_transform(self, field):
            return field

    with pytest.raises(NotImplementedError, match="Field backward transform not implemented"):
        ForwardOnly.reversed()(fl)


def test_pipeline_and_reverse():
    class Add(Transform):
        def __init__(self, k):
            self.k = k

        def forward(self, data):
            return data + self.k

        def backward(self, data):
            return data - self.k

    class Mul(Transform):
        def forward(self, data):
            return data * 2

    p = Add(1) | Add(10)
    assert isinstance(p, Pipeline) and p(0) == 11 and p.backward(11) == 0
    q = p | Mul()  # nested two-element pipelines (R: transform.py:116-131)
    assert isinstance(q.filters[0], Pipeline) and q(0) == 22
    with pytest.raises(NotImplementedError, match="is not reversible"):
        q.backward(1)
    assert repr(Add(1).reverse()).startswith("Reversed(")
    assert Add(1).patch_data_request({"a": 1}) == {"a": 1}


def test_dispatching_filter_contract():
    """R: tests/test_dispatchingfilter.py:44-166."""
    import pandas as pd

    with pytest.raises(TypeError, match="must override at least one"):

        class Nothing(DispatchingFilter):
            pass

    with pytest.raises(TypeError, match="overrides `backward_fields` but not `forward_fields`"):

        class BackOnly(DispatchingFilter):
            def forward_tabular(self, data):
                return data

            def backward_fields(self, data):
                return data

    class Both(DispatchingFilter):
        def forward_fields(self, data):
            return "fields"

        def forward_tabular(self, data):
            return "tabular"

        def backward_fields(self, data):
            return "back"

    f = Both()
    assert f.forward(FieldList([])) == "fields"
    assert f.forward(pd.DataFrame()) == "tabular"
    assert f.backward(FieldList([])) == "back"
    with pytest.raises(TypeError, match="No forward method"):
        f.forward([1, 2])
    with pytest.raises(NotImplementedError, match="No backward method"):
        f.backward(pd.DataFrame())
    with pytest.raises(NotImplementedError):
        f.backward(3)


# ---- fields -----------------------------------------------------------------------------------
def test_array_field_from_dict_is_latitude_major():
    """R: tests/field_filters/test_remove_nans.py:17-45 — flattened order and meshgridded coordinates."""
    f = ArrayField.from_dict(SPECS[0], mars=True)
    assert ArrayField.from_dict(SPECS[0]).metadata(namespace="mars") == {}  # plain dict fields have no MARS namespace
    assert f.shape == (3, 2)
    assert np.array_equal(f.to_numpy(flatten=True), np.arange(6.0))
    lat, lon = f.grid_points()
    assert np.array_equal(lat, [10, 10, 0, 0, -10, -10]) and np.array_equal(lon, [20, 40, 20, 40, 20, 40])
    assert f.metadata("param") == "t" and f.metadata("param", "levelist") == ("t", 500)
    with pytest.raises(KeyError):
        f.metadata("nope")
    assert f.metadata("nope", default=None) is None
    assert f.metadata(namespace="mars") == {"param": "t", "levelist": 500}
    assert f.to_numpy(dtype=np.float32).dtype == np.float32
    assert f.to_numpy(flatten=True, index=[0, 5]).tolist() == [0.0, 5.0]
    assert np.array_equal(f.values, np.arange(6.0))
    assert f.to_numpy() is not f.to_numpy()  # always a copy (R: fields.py:198-199)
    with pytest.raises(NotImplementedError):
        iter(f)


def test_derived_field_metadata_semantics():
    """R: fields.py:468-568."""
    f = ArrayField.from_dict(SPECS[0], mars=True)
    g = new_field_from_numpy(np.ones(6), template=f, param="x", level=None, units=lambda field, key, md: "K")
    assert g.shape == (6,) and g.metadata("param") == "x" and g.metadata("levelist") == 500
    assert g.metadata("level") is None  # an override to None is a value, not "missing"
    assert g.metadata("units") == "K"  # callable overrides are invoked
    md = g.metadata()
    assert md.get("param") == "x" and md["levelist"] == 500 and md.get("zzz", 7) == 7
    assert set(md.keys()) == set(f.metadata().keys())  # template's keys only (R: fields.py:508-509)
    assert g.metadata(namespace="mars") == {"param": "x", "levelist": 500}  # overrides only for present keys
    h = new_field_from_latitudes_longitudes(g, np.array([1.0, 2.0]), np.array([3.0, 4.0]))
    assert h.grid_points()[0].tolist() == [1.0, 2.0] and h.to_latlon()["lon"].tolist() == [3.0, 4.0]
    geo = h.metadata().geography
    assert geo.shape() == (2,) and geo.mars_area() == [2.0, 3.0, 1.0, 4.0] and geo.resolution() == "unknown"
    assert h.metadata("param") == "x"
    c = f.clone(param="y")
    assert c.metadata("param") == "y" and np.array_equal(c.to_numpy(), f.to_numpy())


def test_field_selection():
    """R: tests/test_fields.py:68-134, fields.py:767-797."""
    f = ArrayField.from_dict(SPECS[0])
    assert FieldSelection().match(f)
    assert FieldSelection(param="t").match(f) and not FieldSelection(param="q").match(f)
    assert FieldSelection(param=["q", "t"], levelist=500).match(f)
    assert not FieldSelection(param="t", levelist=[850]).match(f)
    assert FieldSelection(param=None, levelist=[]).match(f)
    g = ArrayField.from_dict({k: v for k, v in SPECS[0].items() if k != "levelist"})
    assert not FieldSelection(levelist=500).match(g)  # KeyError -> False
    with pytest.raises(ValueError, match="Invalid keys"):
        FieldSelection(step=1)
    with pytest.raises(ValueError, match="Invalid value"):
        FieldSelection(param={"a": 1})


def test_fieldlist_surface():
    fl = fieldlist_from_dicts(SPECS)
    assert len(fl) == 2 and fl[1].metadata("param") == "q" and isinstance(fl[0:1], FieldList)
    assert fl.metadata("param") == ["t", "q"]
    assert len(fl.sel(param="q")) == 1 and len(fl.sel(param=["t", "q"], levelist=500)) == 1
    assert fl.to_numpy(flatten=True).shape == (2, 6)
    fl.append(fl[0])
    assert len(fl) == 3


# ---- C ABI ------------------------------------------------------------------------------------
def header_symbols():
    text = open(os.path.join(ROOT, "include", "atx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(atx_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    declared = header_symbols()
    assert len(declared) >= 15
    handle = ctypes.CDLL(native.lib_path())
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/atx.h but not exported"
    assert set(declared) == set(native.SIGNATURES), "native.py must bind exactly the header's entry points"


def test_library_exports_nothing_but_the_header():
    """The dynamic symbol table of libatx.so is EXACTLY the ATX_API declarations of include/atx.h (= native.SIGNATURES): built with
    -fvisibility=hidden and a version script, so no C++ helper, kernel stub or template instance can interpose — or be interposed by —
    another library loaded into the same process (round 5 leaked 1 252 such names)."""
    import shutil
    import subprocess

    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    if not (shutil.which("nm") or os.path.exists(nm)):
        pytest.skip("no nm")
    listing = subprocess.run([nm, "-D", "--defined-only", native.lib_path()], capture_output=True, text=True, check=True).stdout
    defined = sorted(line.split()[-1].split("@")[0] for line in listing.splitlines() if line.strip())
    assert defined == sorted(native.SIGNATURES), sorted(set(defined) ^ set(native.SIGNATURES))[:20]
    assert defined == header_symbols()
    # every one of them a function in the text section, none weak
    kinds = {line.split()[-2] for line in listing.splitlines() if line.strip()}
    assert kinds == {"T"}, kinds
    # the header marks each declaration, and only those, with ATX_API
    text = open(os.path.join(ROOT, "include", "atx.h")).read()
    assert sorted(re.findall(r"^ATX_API [^;(]*?\b(atx_[a-z_0-9]+)\(", text, flags=re.M)) == defined
    # RCCL stays a run-time binding (dlopen in atx_comm.hip), not a link-time dependency the stand-in could not replace
    undefined = subprocess.run([nm, "-D", "--undefined-only", native.lib_path()], capture_output=True, text=True, check=True).stdout
    assert "nccl" not in undefined.lower()


def test_header_is_plain_c_and_links(tmp_path):
    """include/atx.h compiles as C99 with warnings as errors, and a C program linked against libatx.so runs (tests/c_abi/abi_check.c)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    src = os.path.join(ROOT, "tests", "c_abi", "abi_check.c")
    exe = str(tmp_path / "abi_check")
    lib_dir = os.path.dirname(native.lib_path())
    build = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                            "-L", lib_dir, "-latx", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok"), run.stdout + run.stderr


def test_abi_argument_validation_without_a_gpu():
    """Bad arguments are rejected before anything touches HIP, with the reference's exception types."""
    lib = native.load()
    assert lib.atx_version() == 420
    assert lib.atx_strerror(native.ESHAPE) == b"shape mismatch"
    assert lib.atx_regrid_ell(None, None, None, None, 1, 1, 1, 1, 1, 1, 0, 0, 0, None, None, None, 0, None, None) == native.EINVAL
    assert b"null" in lib.atx_last_error()
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    assert lib.atx_regrid_ell(one, one, one, None, 8, 8, 1, 4, 2, 4, 0, 0, 0, None, None, None, 0, None, None) == native.ESHAPE
    assert lib.atx_regrid_ell(one, one, one, None, 8, 8, 3, 4, 4, 4, 0, 0, 0, None, None, None, 0, None, None) == native.EINVAL
    assert lib.atx_regrid_ell(one, one, one, one, 8, 8, 99, 4, 4, 4, 0, 0, 0, None, None, None, 0, None, None) == native.EINVAL
    assert lib.atx_regrid_ell(one, one, one, None, 8, 8, 1, 4, 4, 4, 7, 0, 0, None, None, None, 0, None, None) == native.EINVAL
    assert lib.atx_regrid_ell(one, one, one, None, 8, 8, 1, 4, 4, 4, 0, 0, 1, None, None, None, 0, None, None) == native.EINVAL  # padded needs weights
    assert lib.atx_regrid_ell(one, one, one, None, 8, 8, 1, 4, 4, 4, 0, 0, 0, None, one, None, 0, None, None) == native.EINVAL  # vec_prog without prog
    assert lib.atx_pointwise_stack(one, one, 8, 4, 4, 4, 0, 0, one, None, None, 0, None, None) == native.EINVAL
    assert lib.atx_mask_build(one, 1, one, 8, 99, 0.0, 0, None) == native.EINVAL
    assert lib.atx_relayout(one, one, 8, 4, 4, 8, 0, 1, 0, None) == native.EINVAL  # in place
    assert lib.atx_mask_to_index_workspace(4096 * 3) >= 16
    # a buffer of ZERO elements may have no storage (allocators hand out NULL for empty tensors): validated, nothing done, ATX_OK
    assert lib.atx_relayout(None, None, 0, 4, 4, 0, 0, 1, 0, None) == native.OK
    assert lib.atx_relayout(None, None, 1, 4, 4, 1, 0, 1, 0, None) == native.EINVAL  # ... but one point needs its storage
    assert lib.atx_select_levels(None, None, None, 0, 0, 4, 4, 0, 0, 0, None) == native.OK
    assert lib.atx_pointwise_stack(None, None, 0, 4, 4, 4, 0, 0, one, None, None, 1, None, None) == native.OK
    assert lib.atx_pointwise_stack(None, None, 0, 4, 4, 4, 0, 0, None, None, None, 1, None, None) == native.EINVAL  # the program is never optional
    assert lib.atx_mask_build(None, 1, None, 0, 0, 0.0, 0, None) == native.OK
    assert lib.atx_stream_copy(None, None, 0, None) == native.OK and lib.atx_stream_copy(None, None, 16, None) == native.EINVAL
    assert lib.atx_regrid_ell(one, None, None, None, 8, 0, 1, 4, 4, 4, 0, 0, 0, None, None, None, 0, None, None) == native.OK  # no targets
    assert lib.atx_regrid_ell(one, None, one, None, 8, 1, 1, 4, 4, 4, 0, 0, 0, None, None, None, 0, None, None) == native.EINVAL
    # collective entry points: argument checks come before RCCL or HIP are touched
    assert lib.atx_bcast(None, one, 16, 0, None) == native.EINVAL and b"communicator" in lib.atx_last_error()
    assert lib.atx_comm_init(None, 1, 0, one) == native.EINVAL
    handle = ctypes.c_void_p()
    assert lib.atx_comm_init(ctypes.byref(handle), 2, 5, one) == native.EINVAL and b"rank 5" in lib.atx_last_error()
    assert lib.atx_exchange(None, None, None, None, None, None) == native.EINVAL
    assert lib.atx_all_gather(None, one, one, 16, None) == native.EINVAL
    assert lib.atx_gather_shards(None, one, None, None) == native.EINVAL
    assert lib.atx_comm_destroy(None) == native.OK and lib.atx_comm_rank(None) == native.EINVAL
    assert lib.atx_strerror(native.ECOMM).startswith(b"RCCL")
    with pytest.raises(ValueError, match="128 bytes"):
        native.Comm(1, 0, b"short")
    with pytest.raises(RuntimeError, match="HBM-resident"):
        import torch

        native.regrid_ell(torch.zeros(4, 4), torch.zeros(4, 4), torch.zeros(4, dtype=torch.int32), None, n_src=4, n_tgt=4, k=1,
                          n_lev=4, src_pitch=4, out_pitch=4, layout=native.COLUMNS)


def test_level_program_layout_matches_the_c_struct():
    import torch

    prog = native.level_program([[(native.OP_AFFINE, 0, 2.0, 1.0), (native.OP_COPY, 1, 0.0, 0.0)]], torch.device("cpu"))
    assert prog.numel() == 2 * 24 and native.LEVEL_OP_DTYPE.itemsize == 24
    raw = prog.numpy().tobytes()
    import struct

    assert struct.unpack("<iidd", raw[:24]) == (native.OP_AFFINE, 0, 2.0, 1.0)
    assert struct.unpack("<iidd", raw[24:]) == (native.OP_COPY, 1, 0.0, 0.0)


# ---- CLI: the file formats of make-regrid-file / get-grid ------------------------------------------------
def test_cli_writes_the_reference_file_formats(tmp_path, capsys):
    from anemoi_transform_amd.cli import main

    grid_file, matrix_file, bil_file, mask_file = (str(tmp_path / n) for n in ("grid.npz", "m.npz", "b.npz", "mask.npz"))
    assert main(["get-grid", "o16", "--output", grid_file]) == 0
    g = np.load(grid_file)
    assert set(g) == {"latitudes", "longitudes"} and len(g["latitudes"]) == 4 * 16 * 16 + 36 * 16
    assert main(["make-regrid-file", "knn-matrix", grid_file, "20/20", "--k", "3", "--output", matrix_file]) == 0
    m = np.load(matrix_file)
    assert set(m) == {"matrix_data", "matrix_indices", "matrix_indptr", "matrix_shape", "in_latitudes", "in_longitudes",
                      "out_latitudes", "out_longitudes"}
    assert m["matrix_indices"].dtype == np.int32 and tuple(m["matrix_shape"]) == (10 * 18, len(g["latitudes"]))
    idx, w = interp.knn_inverse_distance(grids.lookup("o16"), grids.lookup("20/20"), k=3)
    assert np.array_equal(m["matrix_indices"].reshape(-1, 3), idx) and np.array_equal(m["matrix_data"].reshape(-1, 3), w)
    assert main(["make-regrid-file", "bilinear-matrix", "o16", "20/20", "--output", bil_file]) == 0
    assert np.allclose(np.load(bil_file)["matrix_data"].reshape(-1, 4).sum(axis=1), 1.0)
    for source in ("f8", "10/10"):  # any row-structured formula grid is a valid source of the bilinear builder
        assert main(["make-regrid-file", "bilinear-matrix", source, "o8", "--output", bil_file]) == 0
        b = np.load(bil_file)
        assert tuple(b["matrix_shape"]) == (len(grids.lookup("o8")["latitudes"]), len(grids.lookup(source)["latitudes"]))
    with pytest.raises(SystemExit):
        main(["make-regrid-file", "bilinear-matrix", grid_file, "o8", "--output", bil_file])  # an npz has no row structure
    lam = str(tmp_path / "lam.npz")
    lat, lon = np.meshgrid(np.linspace(40, 50, 6), np.linspace(0, 10, 6))
    np.savez(lam, latitudes=lat.ravel(), longitudes=lon.ravel())
    assert main(["make-regrid-file", "global-on-lam-mask", "o16", lam, "--distance-km", "400", "--output", mask_file]) == 0
    mask = np.load(mask_file)["mask"]
    assert mask.dtype.kind == "i" and np.all(np.diff(mask) > 0) and len(mask) > 0
    assert main(["filters", "list"]) == 0
    assert "regrid" in capsys.readouterr().out


def test_derived_field_forwards_unknown_attributes(caplog):
    """R: fields.py:69-109 (WrappedField.__getattr__): unknown attributes come from the wrapped field, with a warning;
    `copy` is refused."""
    import logging

    from anemoi_transform_amd.fields import ArrayField, new_field_with_metadata

    base = ArrayField(np.arange(6.0).reshape(3, 2), {"param": "t"}, np.zeros(6), np.zeros(6))
    base.origin = "unit-test"
    derived = new_field_with_metadata(new_field_with_metadata(base, param="u"), levelist=5)
    with caplog.at_level(logging.WARNING):
        assert derived.origin == "unit-test"
    assert "forwarding `origin`" in caplog.text
    with pytest.raises(AttributeError, match="forwarding of `copy` is not supported"):
        derived.copy
    with pytest.raises(AttributeError):
        derived.no_such_attribute
    assert derived.clone(param="v").metadata("param") == "v"


def test_units_and_grid_field_factories():
    """R: fields.py:701-716 (new_field_with_units), :741-759 / :384-466 (new_field_from_grid: anything with ``latlon()``)."""
    from anemoi_transform_amd.fields import ArrayField, new_field_from_grid, new_field_with_units

    base = ArrayField(np.arange(6.0), {"param": "t", "units": "K"}, np.arange(6.0), np.arange(6.0) * 2)
    assert new_field_with_units(base, "degC").metadata("units") == "degC" and base.metadata("units") == "K"

    class Grid:
        def latlon(self):
            return np.linspace(10, 60, 6), np.linspace(-20, 30, 6)

    moved = new_field_from_grid(base, Grid())
    lat, lon = moved.grid_points()
    assert np.array_equal(lat, np.linspace(10, 60, 6)) and np.array_equal(lon, np.linspace(-20, 30, 6))
    assert np.array_equal(moved.to_numpy(), base.to_numpy()) and moved.metadata("param") == "t"
    assert moved.to_latlon()["lat"][0] == 10.0


def test_vector_program_host_helper():
    """atx_vector_program (host only): per-vector view of a per-level program; levels that differ mark the vector MIXED,
    parameters are compared as the kernel will see them (rounded to the stack's dtype), padding levels join any operator."""
    lib = native.load()
    ops = [(native.OP_AFFINE, 0, 2.0, 1.0)] * 5 + [(native.OP_MUL, 0, 9.80665, 0.0)] * 3 + [(native.OP_AFFINE, 1, 2.0, 1.0)]
    ops[3] = (native.OP_AFFINE, 0, 2.0 + 1e-12, 1.0)  # equal to 2.0 in float32, different in float64
    host = np.zeros(len(ops), dtype=native.LEVEL_OP_DTYPE)
    for i, o in enumerate(ops):
        host[i] = o
    for code, vec, want in ((native.F32, 4, [native.OP_AFFINE, -1, native.OP_AFFINE]),
                            (native.F64, 2, [native.OP_AFFINE, -1, -1, native.OP_MUL, native.OP_AFFINE])):
        n = lib.atx_vector_program(host.ctypes.data, 1, len(ops), code, None, 0)
        C = (len(ops) + vec - 1) // vec
        assert C == len(want)
        # second part, from the next 16-byte boundary: p0[Lp], p1[Lp] in the stack's type and one code byte per level
        np_t, Lp, start = (np.float32, np.float64)[code == native.F64], C * vec, (C * 24 + 15) // 16 * 16
        assert n == -(-(start + Lp * (2 * np.dtype(np_t).itemsize + 1)) // 24)
        out = np.zeros(n, dtype=native.LEVEL_OP_DTYPE)
        assert lib.atx_vector_program(host.ctypes.data, 1, len(ops), code, out.ctypes.data, n) == n
        assert out["op"][:C].tolist() == want
        assert out["use_mask"][C - 1] == 1  # the last (partial) vector holds only level 8
        raw = out.view(np.uint8)
        padded = ops + [ops[-1]] * (Lp - len(ops))  # the padding of the last vector repeats the last level
        p0 = raw[start:start + Lp * np.dtype(np_t).itemsize].view(np_t)
        p1 = raw[start + Lp * np.dtype(np_t).itemsize:start + 2 * Lp * np.dtype(np_t).itemsize].view(np_t)
        codes = raw[start + 2 * Lp * np.dtype(np_t).itemsize:][:Lp]
        assert p0.tolist() == [np_t(o[2]) for o in padded] and p1.tolist() == [np_t(o[3]) for o in padded]
        assert codes.tolist() == [o[0] | (o[1] << 7) for o in padded]
        # the capacity of `out` is checked: one entry short is refused and NOTHING is written (0.3 had no such argument)
        guard = np.full(n, 7, dtype=np.uint8).repeat(24).view(native.LEVEL_OP_DTYPE)
        before = guard.tobytes()
        assert lib.atx_vector_program(host.ctypes.data, 1, len(ops), code, guard.ctypes.data, n - 1) == native.EWORKSPACE
        assert guard.tobytes() == before and b"needs" in lib.atx_last_error()
    assert lib.atx_vector_program(None, 1, 4, native.F32, None, 0) == native.EINVAL
    assert lib.atx_vector_program(host.ctypes.data, 9, 1, native.F32, None, 0) == native.EINVAL


def test_shard_bounds_invariants_on_random_plans(monkeypatch):
    """Whatever the index table, world size and operator shape: bounds are monotone, start at 0, end at n_tgt; the shards
    are contiguous row ranges of the plan and, applied one by one, reproduce the unsharded result bit for bit (kernels: the
    oracle-backed double)."""
    import native_double
    import torch
    from anemoi_transform_amd.stack import Stack

    native_double.install(monkeypatch)
    rng = np.random.default_rng(77)
    for case in range(25):
        n_src, n_tgt = int(rng.integers(1, 400)), int(rng.integers(0, 300))
        world = int(rng.integers(1, 10))
        kind = rng.choice(["gather", "ell", "padded", "csr"])
        if kind == "gather":
            plan = GatherPlan(n_src, n_tgt, index=rng.integers(0, n_src, n_tgt))
        elif kind == "ell":
            k = int(rng.integers(1, 6))
            plan = GatherPlan(n_src, n_tgt, index=rng.integers(0, n_src, (n_tgt, k)), weights=rng.random((n_tgt, k)))
        else:
            lengths = rng.integers(0, 5 if kind == "padded" else 12, n_tgt)
            indptr = np.concatenate([[0], np.cumsum(lengths)])
            matrix = dict(matrix_data=rng.random(int(indptr[-1])), matrix_indices=rng.integers(0, n_src, int(indptr[-1])).astype(np.int32),
                          matrix_indptr=indptr.astype(np.int32), matrix_shape=(n_tgt, n_src))
            plan = GatherPlan.from_matrix(matrix) if n_tgt else GatherPlan(n_src, 0, index=np.zeros((0, 1), dtype=np.int64))
        b = plan.bounds(world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == n_tgt and all(x <= y for x, y in zip(b, b[1:])), (case, b)
        assert [plan.shard_range(r, world) for r in range(world)] == list(zip(b[:-1], b[1:]))
        x = Stack.from_fields(rng.standard_normal((3, n_src)), dev=torch.device("cpu"))
        full = plan.apply(x).numpy()
        parts = [plan.shard(r, world).apply(x).numpy() for r in range(world)]
        assert np.array_equal(np.concatenate(parts, axis=1), full, equal_nan=True), (case, kind, world)
        # the same with another weight of the cost model (bench.py cuts its one-short-launch-per-rank step with TARGET_COST_SHORT_LAUNCH):
        # its own cut, remembered beside the default one, same invariants, same bits
        heavy = plan.bounds(world, target_cost=TARGET_COST_SHORT_LAUNCH)
        assert len(heavy) == world + 1 and heavy[0] == 0 and heavy[-1] == n_tgt and all(x_ <= y_ for x_, y_ in zip(heavy, heavy[1:]))
        assert plan.bounds(world) == b and plan.bounds(world, target_cost=TARGET_COST) == b  # the default cut is untouched
        assert [plan.shard_range(r, world, TARGET_COST_SHORT_LAUNCH) for r in range(world)] == list(zip(heavy[:-1], heavy[1:]))
        parts = [plan.shard(r, world, TARGET_COST_SHORT_LAUNCH).apply(x).numpy() for r in range(world)]
        assert [p.shape[1] for p in parts] == [hi - lo for lo, hi in zip(heavy[:-1], heavy[1:])]
        assert np.array_equal(np.concatenate(parts, axis=1), full, equal_nan=True), (case, kind, world)


def test_a_heavier_target_weight_moves_targets_away_from_the_shards_that_share_sources():
    """On a lat-lon target the polar shards read few source columns for many targets: the heavier a target weighs in the cost model,
    the fewer targets they get (the cut bench.py uses for its strong-scaling step; gather.TARGET_COST_SHORT_LAUNCH)."""
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.interp import knn_inverse_distance

    src, tgt = lookup("o48"), lookup([2.0, 2.0])
    idx, w = knn_inverse_distance(src, tgt, k=4)
    plan = GatherPlan(len(src["latitudes"]), len(tgt["latitudes"]), index=idx, weights=w)
    light, heavy = plan.bounds(8, target_cost=0.2), plan.bounds(8, target_cost=3.0)
    size = lambda b, r: b[r + 1] - b[r]  # noqa: E731
    assert size(heavy, 0) < size(light, 0) and size(heavy, 7) < size(light, 7)  # polar shards shrink
    assert size(heavy, 3) > size(light, 3) and size(heavy, 4) > size(light, 4)  # equatorial ones grow
    default = plan.bounds(8)
    assert size(light, 0) > size(default, 0) > size(plan.bounds(8, target_cost=TARGET_COST_SHORT_LAUNCH), 0) > size(heavy, 0)


def test_widths_between_the_compile_time_forms_are_padded(monkeypatch):
    """k = 9-11 / 13-15: GatherPlan pads the table to 12 / 16 with absent entries (index -1, skipped in the sum) so that the direct
    kernel's compile-time form runs it — the values are those of the k given."""
    import native_double
    import torch
    from anemoi_transform_amd.stack import Stack
    from oracle import oracle

    native_double.install(monkeypatch)
    rng = np.random.default_rng(9)
    n_src, n_tgt = 300, 120
    x = Stack.from_fields(rng.standard_normal((4, n_src)), dev=torch.device("cpu"))
    for k, wide in ((9, 12), (11, 12), (12, 12), (13, 16), (15, 16), (16, 16), (8, 8), (17, 17)):
        idx, w = rng.integers(0, n_src, (n_tgt, k)), rng.random((n_tgt, k))
        plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
        assert plan.k == wide and plan.padded == (wide != k)
        want = np.stack([oracle.csr_apply(w.reshape(-1), idx.reshape(-1), np.arange(n_tgt + 1) * k, (n_tgt, n_src), f) for f in x.numpy()])
        assert np.array_equal(plan.apply(x).numpy(), want)
        assert np.array_equal(np.concatenate([plan.shard(r, 3).apply(x).numpy() for r in range(3)], axis=1), want)


def test_long_rows_on_field_major_stacks_go_through_columns(monkeypatch):
    """A field-major stack and rows of more than 8 entries: GatherPlan.apply converts to columns, gathers and converts back (4x faster
    than the field-major gather on MI355X) — same values, same layout as what came in; short rows stay on the field-major kernel."""
    import native_double
    import torch
    from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack

    native_double.install(monkeypatch)
    seen = []
    real = native.relayout
    monkeypatch.setattr(native, "relayout", lambda *a, **k: (seen.append((k["src_layout"], k["dst_layout"])), real(*a, **k))[1])
    rng = np.random.default_rng(3)
    n_src, n_tgt = 200, 90
    fields = rng.standard_normal((5, n_src))
    x_f = Stack.from_fields(fields, dev=torch.device("cpu")).to_layout(FIELDS)
    seen.clear()
    for k, converts in ((4, False), (12, True)):
        plan = GatherPlan(n_src, n_tgt, index=rng.integers(0, n_src, (n_tgt, k)), weights=rng.random((n_tgt, k)))
        out = plan.apply(x_f)
        assert out.layout == FIELDS and seen == ([(FIELDS, COLUMNS), (COLUMNS, FIELDS)] if converts else []), (k, seen)
        assert np.array_equal(out.numpy(), plan.apply(x_f.to_layout(COLUMNS)).numpy())
        seen.clear()
    lengths = rng.integers(6, 20, n_tgt)
    indptr = np.concatenate([[0], np.cumsum(lengths)])
    csr = GatherPlan(n_src, n_tgt, csr=(rng.random(int(indptr[-1])), rng.integers(0, n_src, int(indptr[-1])).astype(np.int32), indptr.astype(np.int32)))
    out = csr.apply(x_f)
    assert out.layout == FIELDS and seen[0] == (FIELDS, COLUMNS) and seen[-1] == (COLUMNS, FIELDS)


def test_gather_into_a_caller_kept_stack_and_bound_launches(monkeypatch):
    """`GatherPlan.apply(out=)` writes into a stack the caller keeps (no allocation per call) and refuses one of the wrong shape;
    `GatherPlan.bind` gives a launch that repeats the gather on the same buffers and sees new CONTENTS of the source."""
    import native_double
    import torch
    from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack
    from oracle import oracle

    native_double.install(monkeypatch)
    rng = np.random.default_rng(4)
    n_src, n_tgt, k = 150, 64, 4
    fields = rng.standard_normal((3, n_src))
    idx, w = rng.integers(0, n_src, (n_tgt, k)), rng.random((n_tgt, k))
    want = lambda f: np.stack([oracle.csr_apply(w.reshape(-1), idx.reshape(-1), np.arange(n_tgt + 1) * k, (n_tgt, n_src), x) for x in f])  # noqa: E731
    for layout in (COLUMNS, FIELDS):
        x = Stack.from_fields(fields, dev=torch.device("cpu"), layout=layout)
        for plan in (GatherPlan(n_src, n_tgt, index=idx, weights=w),
                     GatherPlan(n_src, n_tgt, csr=(w.reshape(-1), idx.reshape(-1).astype(np.int32), (np.arange(n_tgt + 1) * k).astype(np.int32)))):
            out = x.new_like(n_pts=n_tgt)
            assert plan.apply(x, out=out) is out and np.array_equal(out.numpy(), want(fields))
            with pytest.raises(ValueError, match="out= must be"):
                plan.apply(x, out=x.new_like(n_pts=n_tgt + 1))
            with pytest.raises(ValueError, match="out= must be"):
                plan.apply(x, out=Stack.empty(n_tgt, 3, torch.float32, torch.device("cpu"), layout))
            launch, kept = plan.bind(x)
            launch()
            assert np.array_equal(kept.numpy(), want(fields))
            x.data.mul_(2.0)  # new contents, same storage
            launch()
            assert np.array_equal(kept.numpy(), want(2.0 * fields))
            x.data.mul_(0.5)
    # a field-major stack and long rows: apply(out=) still lands in the caller's stack (through columns and back)
    wide = GatherPlan(n_src, n_tgt, index=rng.integers(0, n_src, (n_tgt, 12)), weights=rng.random((n_tgt, 12)))
    x_f = Stack.from_fields(fields, dev=torch.device("cpu"), layout=FIELDS)
    out = x_f.new_like(n_pts=n_tgt)
    assert wide.apply(x_f, out=out) is out and np.array_equal(out.numpy(), wide.apply(x_f).numpy())


def test_no_streaming_kernel_uses_scratch_memory():
    """Read from the code objects inside libatx.so (tools/kernel_resources.py; no GPU): only the k-NN traversal (its per-lane stack) and
    rocPRIM's radix sort may use private scratch memory.  A private array indexed with a run-time subscript lands there — the field-major
    per-point kernel ran at 0.41 of the HBM peak instead of 0.83 for two rounds because of one."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        import kernel_resources
    finally:
        sys.path.pop(0)
    rows = kernel_resources.kernel_resources(native.lib_path())
    names = kernel_resources.demangle([r["name"] for r in rows])
    assert len(rows) > 300  # every translation unit was read
    offenders = [n for r, n in zip(rows, names) if r["scratch"] > 0 and "knn_query_kernel" not in n and "rocprim" not in n]
    assert offenders == []
    assert any("pointwise_fields_rows_kernel" in n for n in names) and any("regrid_cols_ell_direct_kernel" in n for n in names)


def test_design_tables_are_generated():
    """DESIGN.md's measured tables come from the tracked JSON records (tools/design_tables.py): hand-copied cells drifted in round 2."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "design_tables.py"), "--check"], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr


def test_row_split_through_the_float_reciprocal_is_exact():
    """atx_combine.hip splits a lane's vector index into (row, column) with floor((off + 0.5) * rcp(vec_per_row)) in float32 instead of an
    integer division (off < vec_per_row + 512, vec_per_row < 2^20).  The same arithmetic in numpy float32, with the reciprocal one ulp low,
    exact and one ulp high (the hardware's v_rcp_f32 is good to an ulp): equal to the integer quotient for every offset of 6 000 divisors —
    all of 1 .. 2048, the column counts of the stacks in use, random ones up to 2^20."""
    rng = np.random.default_rng(0)
    divisors = np.unique(np.concatenate([np.arange(1, 2049), rng.integers(2049, 1 << 20, 4000), [35, 69, 138, 206, (1 << 20) - 1]]))
    for vpr in divisors:
        off = np.arange(0, vpr + 513, dtype=np.int64)
        if off.size > 200_000:  # long rows: both ends, the neighbourhood of the row boundary and a random middle
            off = np.unique(np.concatenate([off[:50_000], off[-50_000:], vpr + np.arange(-600, 513), rng.integers(0, vpr + 513, 50_000)]))
        want = off // vpr
        inv0 = np.float32(1.0) / np.float32(vpr)
        for inv in (np.nextafter(inv0, np.float32(0)), inv0, np.nextafter(inv0, np.float32(1))):
            got = ((off.astype(np.float32) + np.float32(0.5)) * inv).astype(np.int64)
            assert np.array_equal(got, want), (int(vpr), float(inv))
