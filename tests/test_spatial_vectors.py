"""Index / mask builders against vectors produced by RUNNING the reference's own ``spatial.py``
(``tests/golden/spatial_vectors.npz``, written by ``tests/golden/make_spatial_vectors.py`` in the build container).

Two things are held to the file here, on the CPU: the oracle's restatement (``oracle/oracle.py``) and the package's host
builders (``anemoi_transform_amd.spatial`` / ``interp``).  The device builders are held to it in
``tests/test_gpu_spatial_vectors.py``.  Everything is index / boolean work or float64 chord distances: ``array_equal``.
"""

from __future__ import annotations

import hashlib
import json
import os

import numpy as np
import pytest

from anemoi_transform_amd import interp, spatial
from anemoi_transform_amd.grids import lookup
from oracle import oracle

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spatial_vectors.npz")


class Vectors:
    """The fixture file plus the formula grids it names (regenerated here and checked against the recorded hashes)."""

    def __init__(self):
        self.file = np.load(FIXTURE)
        self.manifest = json.loads(str(self.file["manifest"]))
        self._grids: dict[str, dict] = {}

    def grid(self, name: str) -> dict:
        if name not in self._grids:
            entry = self.manifest["grids"][name]
            g = lookup(entry["spec"])
            h = hashlib.sha256()
            for key in ("latitudes", "longitudes"):
                h.update(np.ascontiguousarray(g[key], dtype=np.float64).tobytes())
            assert h.hexdigest() == entry["sha256"], f"grid {name} is not the one the vectors were generated on"
            assert len(g["latitudes"]) == entry["n"]
            self._grids[name] = g
        return self._grids[name]

    def lam(self, name: str):
        return self.file[f"lam/{name}/lats"], self.file[f"lam/{name}/lons"]

    def mask(self, case: dict) -> np.ndarray:
        return np.unpackbits(self.file[case["key"]])[: case["n"]].astype(bool)


VECTORS = Vectors()
MANIFEST = VECTORS.manifest


def ids(cases):
    return [c["key"] for c in cases]


def test_the_file_says_where_it_comes_from():
    assert MANIFEST["generator"] == "tests/golden/make_spatial_vectors.py"
    assert "spatial.py" in MANIFEST["reference_module"]
    # the two constants the limited-area builders use are what the package and the oracle carry (R: constants.py:11-26)
    assert MANIFEST["constants"]["R_earth_km"] == spatial.R_earth_km == oracle.R_EARTH_KM
    assert MANIFEST["constants"]["radian"] == spatial.radian == oracle.RADIAN
    assert len(MANIFEST["nearest_grid_points"]) == 12 and len(MANIFEST["cutout_mask"]) >= 18
    # the tie-heavy pair really has ties, the bounded variants really miss neighbours
    by_key = {c["key"]: c for c in MANIFEST["nearest_grid_points"]}
    assert by_key["ngp/reg10_to_5deg/k4/unbounded"]["rows_with_equal_adjacent_distances"] > 500
    assert all(c["missing"] > 0 for c in MANIFEST["nearest_grid_points"] if c["max_distance"] is not None)


# ---- nearest_grid_points (R: spatial.py:587-635) ------------------------------------------------------------------------------
IMPLS_NGP = [pytest.param(oracle.nearest_grid_points, id="oracle"), pytest.param(interp.nearest_grid_points, id="package-host")]


@pytest.mark.parametrize("nearest_grid_points", IMPLS_NGP)
@pytest.mark.parametrize("case", MANIFEST["nearest_grid_points"], ids=ids(MANIFEST["nearest_grid_points"]))
def test_nearest_grid_points(case, nearest_grid_points, monkeypatch):
    monkeypatch.setenv("ATX_KNN", "ckdtree")
    interp.knn_cache_clear()
    src, tgt = VECTORS.grid(case["source"]), VECTORS.grid(case["target"])
    want_idx, want_dist = VECTORS.file[case["key"] + "/idx"], VECTORS.file[case["key"] + "/dist"]
    idx, dist = nearest_grid_points(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"],
                                    max_distance=case["max_distance"], num_neighbours_to_return=case["k"], return_distances=True)
    assert list(idx.shape) == case["shape"] and idx.dtype.kind == "i"
    assert np.array_equal(idx, want_idx)
    assert np.array_equal(dist, want_dist)  # float64 chord distances, inf where the bound leaves no neighbour
    only = nearest_grid_points(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"],
                               max_distance=case["max_distance"], num_neighbours_to_return=case["k"])
    assert np.array_equal(only, want_idx)  # indices alone when distances are not asked for (R: spatial.py:633-635)


# ---- cutout_mask (R: spatial.py:294-440) -----------------------------------------------------------------------------------
IMPLS_CUTOUT = [pytest.param(oracle.cutout_mask, id="oracle"), pytest.param(spatial.cutout_mask, id="package-host")]


@pytest.mark.parametrize("cutout_mask", IMPLS_CUTOUT)
@pytest.mark.parametrize("case", MANIFEST["cutout_mask"], ids=ids(MANIFEST["cutout_mask"]))
def test_cutout_mask(case, cutout_mask):
    lats, lons = VECTORS.lam(case["lam"])
    glob = VECTORS.grid(case["global"])
    mask = cutout_mask(lats, lons, glob["latitudes"].copy(), glob["longitudes"].copy(), **case["options"])
    want = VECTORS.mask(case)
    assert mask.dtype == bool and mask.shape == want.shape
    assert int(want.sum()) == case["kept"]
    assert np.array_equal(mask, want)


@pytest.mark.parametrize("cutout_mask", IMPLS_CUTOUT)
@pytest.mark.parametrize("case", MANIFEST["cutout_mask_errors"], ids=[json.dumps(c["options"]) for c in MANIFEST["cutout_mask_errors"]])
def test_cutout_mask_refuses_what_the_reference_refuses(case, cutout_mask):
    """A str distance reaches ``_distance_km_to_resolution`` in ``global_on_lam_mask`` but not here: the reference's own
    ``min_distance_km >= 0.0`` assertion raises TypeError for it (R: spatial.py:336)."""
    lats, lons = VECTORS.lam(case["lam"])
    glob = VECTORS.grid(case["global"])
    error = {"TypeError": TypeError, "AssertionError": AssertionError}[case["raises"]]
    with pytest.raises(error):
        cutout_mask(lats, lons, glob["latitudes"], glob["longitudes"], **case["options"])


# ---- thinning_mask, global_on_lam_mask, cropping_mask ----------------------------------------------------------------------
@pytest.mark.parametrize("thinning_mask", [pytest.param(oracle.thinning_mask, id="oracle"), pytest.param(spatial.thinning_mask, id="package-host")])
@pytest.mark.parametrize("case", MANIFEST["thinning_mask"], ids=ids(MANIFEST["thinning_mask"]))
def test_thinning_mask(case, thinning_mask):
    lats, lons = VECTORS.lam(case["lam"])
    glob = VECTORS.grid(case["global"])
    indices = thinning_mask(lats, lons, glob["latitudes"], glob["longitudes"], cropping_distance=case["cropping_distance"])
    assert len(indices) == case["n"]
    assert np.array_equal(indices, VECTORS.file[case["key"]])


@pytest.mark.parametrize("global_on_lam_mask",
                         [pytest.param(oracle.global_on_lam_mask, id="oracle"), pytest.param(spatial.global_on_lam_mask, id="package-host")])
@pytest.mark.parametrize("case", MANIFEST["global_on_lam_mask"], ids=ids(MANIFEST["global_on_lam_mask"]))
def test_global_on_lam_mask(case, global_on_lam_mask):
    lats, lons = VECTORS.lam(case["lam"])
    glob = VECTORS.grid(case["global"])
    indices = global_on_lam_mask(lats, lons, glob["latitudes"], glob["longitudes"], distance_km=case["distance_km"])
    want = VECTORS.file[case["key"]]
    assert len(want) == case["n"] and np.all(np.diff(want) > 0)  # sorted unique: what regrid(mask=...) consumes (R: spatial.py:533-536)
    assert np.array_equal(indices, want)


@pytest.mark.parametrize("cropping_mask", [pytest.param(oracle.cropping_mask, id="oracle"), pytest.param(spatial.cropping_mask, id="package-host")])
@pytest.mark.parametrize("case", MANIFEST["cropping_mask"], ids=ids(MANIFEST["cropping_mask"]))
def test_cropping_mask(case, cropping_mask):
    g = VECTORS.grid(case["grid"])
    lons = g["longitudes"] if case["longitudes"] == "0..360" else np.where(g["longitudes"] >= 180.0, g["longitudes"] - 360.0, g["longitudes"])
    mask = cropping_mask(g["latitudes"], lons, *case["box"])
    want = VECTORS.mask(case)
    assert int(want.sum()) == case["inside"] > 0
    assert np.array_equal(mask, want)


def test_a_regrid_mask_built_from_the_vectors_selects_those_points():
    """The index list of ``global_on_lam_mask`` is what ``regrid(mask=...)`` gathers with (R: regrid.py:404-429): the oracle's
    ``x[..., mask]`` on the reference-produced indices picks exactly the global points within the distance."""
    case = next(c for c in MANIFEST["global_on_lam_mask"] if c["lam"] == "europe" and c["distance_km"] == 60.0)
    glob = VECTORS.grid(case["global"])
    indices = VECTORS.file[case["key"]].astype(np.int64)
    picked = oracle.masked_subset(glob["latitudes"], indices)
    assert picked.shape == (case["n"],) and picked.min() > 33.0 and picked.max() < 57.0


# ---- the two regrid statements on the reference's own index tables (R: regrid.py:380, :310) -------------------------------------
def regrid_case_inputs(case):
    """(source fields, k = 1 indices, k = 4 indices, k = 4 inverse-distance weights) of a `regrid` case: the field is regenerated from its seed,
    the tables are the reference-run `nearest_grid_points` vectors of the same file."""
    src = VECTORS.grid(case["source"] if isinstance(case["source"], str) else "x".join(str(v) for v in case["source"]))
    rng = np.random.default_rng(case["seed"])
    lat, lon = np.deg2rad(src["latitudes"]), np.deg2rad(src["longitudes"])
    fields = np.stack([280.0 + 30.0 * np.sin(lat) * np.cos(2.0 * lon + 0.1 * l) + rng.standard_normal(len(lat)) for l in range(case["levels"])])
    idx1 = VECTORS.file[f"ngp/{case['pair']}/k1/unbounded/idx"].astype(np.int64)
    idx4 = VECTORS.file[f"ngp/{case['pair']}/k4/unbounded/idx"].astype(np.int64)
    inv = 1.0 / np.maximum(VECTORS.file[f"ngp/{case['pair']}/k4/unbounded/dist"], 1e-12)
    return fields, idx1, idx4, inv / inv.sum(axis=1, keepdims=True)


@pytest.mark.parametrize("case", MANIFEST["regrid"], ids=ids(MANIFEST["regrid"]))
def test_regrid_statements_on_reference_tables(case):
    """The oracle's two regrid statements reproduce the recorded outputs (the same numpy / scipy calls on the reference's own tables)."""
    fields, idx1, idx4, weights = regrid_case_inputs(case)
    n_tgt, n_src = idx4.shape[0], fields.shape[1]
    assert np.array_equal(oracle.gather_nn(fields, idx1), VECTORS.file[case["key"] + "/nearest"])
    indptr = np.arange(n_tgt + 1) * 4
    got = np.stack([oracle.csr_apply(weights.reshape(-1), idx4.reshape(-1), indptr, (n_tgt, n_src), f) for f in fields])
    assert np.array_equal(got, VECTORS.file[case["key"] + "/knn4"])
