"""Mask / index builders: the reference's known answers (R: tests/test_spatial.py) pin BOTH the
oracle's loop restatement and the package's vectorised builders; random cases tie the two together."""

from __future__ import annotations

import numpy as np
import pytest

from anemoi_transform_amd import spatial
from oracle import oracle

IMPLS = [pytest.param(oracle.cutout_mask, id="oracle"), pytest.param(spatial.cutout_mask, id="package")]


def lam_grid(lat0, lat1, lon0, lon1, n):
    lats, lons = np.meshgrid(np.linspace(lat0, lat1, n), np.linspace(lon0, lon1, n))
    return lats.flatten(), lons.flatten()


@pytest.mark.parametrize("cutout_mask", IMPLS)
@pytest.mark.parametrize("cropping_distance", [1.0, 3.0, 5.0])
def test_cutout_mask_with_max_distance(cutout_mask, cropping_distance):
    """R: tests/test_spatial.py:20-50 — known answer [True, False, False, False, False, False]."""
    lam_lats, lam_lons = lam_grid(44.0, 46.0, 0.0, 2.0, 11)
    global_lats = np.array([43.1, 44.0, 45.0, 45.5, 46.0, 50.0])
    global_lons = np.array([359.1, 359.5, 0.0, 1.0, 2.0, 0.0])
    mask = cutout_mask(lam_lats, lam_lons, global_lats, global_lons, cropping_distance=cropping_distance, max_distance_km=250.0)
    assert isinstance(mask, np.ndarray) and mask.shape == global_lats.shape
    assert np.array_equal(mask, np.array([True, False, False, False, False, False]))


@pytest.mark.parametrize("cutout_mask", IMPLS)
def test_cutout_mask_with_min_distance(cutout_mask):
    """R: tests/test_spatial.py:53-79 — known answer [False, False, False, False, True]."""
    lam_lats, lam_lons = lam_grid(44.0, 46.0, 0.0, 2.0, 11)
    global_lats = np.array([44.0, 45.0, 46.0, 46.1, 47.5])
    global_lons = np.array([0.0, 1.0, 2.0, -0.1, -1.5])
    mask = cutout_mask(lam_lats, lam_lons, global_lats, global_lons, min_distance_km=100.0)
    assert np.array_equal(mask, np.array([False, False, False, False, True]))


@pytest.mark.parametrize("cutout_mask", IMPLS)
def test_cutout_mask_array_shapes(cutout_mask):
    """R: tests/test_spatial.py:82-92."""
    with pytest.raises(AssertionError):
        cutout_mask(np.array([[45.0, 45.0], [46.0, 46.0]]), np.array([[0.0, 1.0], [0.0, 1.0]]), np.array([45.0]), np.array([0.0]))


@pytest.mark.parametrize("cutout_mask", IMPLS)
def test_cutout_mask_large_grid(cutout_mask):
    """R: tests/test_spatial.py:115-145."""
    lam_lats, lam_lons = lam_grid(40.0, 50.0, 0.0, 10.0, 21)
    global_lats, global_lons = lam_grid(30.0, 60.0, -10.0, 20.0, 31)
    for kw in (dict(max_distance_km=100), dict(max_distance_km=100.0)):  # R: tests/test_spatial.py:95-112
        small_lats, small_lons = lam_grid(44.0, 46.0, 0.0, 2.0, 11)
        assert isinstance(cutout_mask(small_lats, small_lons, np.array([45.0, 46.0]), np.array([0.0, 2.0]), **kw), np.ndarray)
    mask = cutout_mask(lam_lats, lam_lons, global_lats, global_lons, min_distance_km=150.0, max_distance_km=300.0)
    assert mask.shape == (961,) and mask.dtype == bool and np.any(mask) and not np.all(mask)


@pytest.mark.parametrize("seed", range(4))
def test_vectorised_builders_equal_the_loop_restatement(seed):
    rng = np.random.default_rng(seed)
    lam_lats, lam_lons = lam_grid(35.0 + seed, 55.0, -5.0, 15.0 + seed, 17)
    lam_lats = lam_lats + rng.normal(0, 0.05, lam_lats.shape)
    global_lats = rng.uniform(20, 70, 1500)
    global_lons = rng.uniform(-30, 40, 1500) % 360
    for kw in (dict(), dict(min_distance_km=80.0), dict(max_distance_km=400.0), dict(min_distance_km=20, neighbours=4)):
        a = spatial.cutout_mask(lam_lats, lam_lons, global_lats, global_lons, **kw)
        b = oracle.cutout_mask(lam_lats, lam_lons, global_lats, global_lons, **kw)
        assert np.array_equal(a, b), kw
    assert np.array_equal(spatial.thinning_mask(lam_lats, lam_lons, global_lats, global_lons),
                          oracle.thinning_mask(lam_lats, lam_lons, global_lats, global_lons))
    a = spatial.global_on_lam_mask(lam_lats, lam_lons, global_lats, global_lons, distance_km=60.0)
    assert np.array_equal(a, oracle.global_on_lam_mask(lam_lats, lam_lons, global_lats, global_lons, distance_km=60.0))
    assert np.all(np.diff(a) > 0)  # sorted unique: what regrid(mask=...) consumes (R: spatial.py:533-536)
    box = spatial.cropping_mask(global_lats, global_lons, 50, 350, 30, 370)
    assert np.array_equal(box, oracle.cropping_mask(global_lats, global_lons, 50, 350, 30, 370)) and box.any()


# ---- device k-NN wrapper: tie handling (host logic, through the CPU double of the kernel) --------------------------
def _regular(dlat, dlon):
    from anemoi_transform_amd.grids import regular_latlon_grid

    return regular_latlon_grid(dlat, dlon)


@pytest.mark.parametrize("k", [1, 3, 4])
def test_device_knn_wrapper_settles_ties_like_ckdtree(monkeypatch, k):
    """``interp.device_knn`` asks the kernel for k+1 neighbours, finds rows with exactly equidistant candidates and lets
    cKDTree decide them: the table must EQUAL the reference statement's (R: spatial.py:628) — on a tie-heavy case, a
    lat-lon source whose pole rows are dozens of coincident points, queried from a symmetric lat-lon target."""
    import native_double
    from anemoi_transform_amd import interp

    native_double.install(monkeypatch)
    src, tgt = _regular(15.0, 15.0), _regular(10.0, 10.0)
    want_i, want_d = oracle.nearest_grid_points(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"],
                                                num_neighbours_to_return=k, return_distances=True)
    got_i, got_d = interp.nearest_grid_points_device(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"],
                                                     num_neighbours_to_return=k, return_distances=True)
    assert np.array_equal(got_i, want_i) and np.array_equal(got_d, want_d)
    # the kernel's own order (lower index first) differs from cKDTree's on this case — the wrapper really had work to do
    raw_i = interp.nearest_grid_points_device(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"],
                                              num_neighbours_to_return=k, ties="index")
    if k > 1:  # (for k = 1 cKDTree happens to pick the lower index on this case too)
        assert not np.array_equal(raw_i, want_i)
    _, _, resolved = interp.device_knn(interp.unit_sphere_xyz(src["latitudes"], src["longitudes"]),
                                       interp.unit_sphere_xyz(tgt["latitudes"], tgt["longitudes"]), k)
    assert resolved <= len(tgt["latitudes"]) and (resolved > 0 or k == 1)


def test_device_knn_wrapper_max_distance_and_arguments(monkeypatch):
    import native_double
    from anemoi_transform_amd import interp

    native_double.install(monkeypatch)
    src, tgt = _regular(30.0, 30.0), _regular(20.0, 20.0)
    for k in (1, 2):
        want = oracle.nearest_grid_points(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"], max_distance=0.2,
                                          num_neighbours_to_return=k, return_distances=True)
        got = interp.nearest_grid_points_device(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"],
                                                max_distance=0.2, num_neighbours_to_return=k, return_distances=True)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert (want[0] == len(src["latitudes"])).any()  # some targets have no source within reach
    xyz = interp.unit_sphere_xyz(src["latitudes"], src["longitudes"])
    with pytest.raises(ValueError):
        interp.device_knn(xyz, xyz, 17)
    with pytest.raises(ValueError):
        interp.device_knn(xyz, xyz, 2, ties="random")
    i, d2, n = interp.device_knn(xyz, xyz[:0], 2)
    assert i.shape == (0, 2) and d2.shape == (0, 2) and n == 0


def test_cutout_mask_device_argument_errors_match_the_host_path(monkeypatch):
    """``neighbours`` beyond the number of LAM points must fail as the host path does (IndexError from
    indexing with cKDTree's padding index), never reach the kernel; beyond the kernel's 16 it is a ValueError."""
    import native_double
    from anemoi_transform_amd import spatial

    native_double.install(monkeypatch)
    lam_lats, lam_lons = np.array([45.0, 45.5, 46.0]), np.array([1.0, 2.0, 1.5])
    g_lats, g_lons = np.array([45.4, 50.0, 44.0]), np.array([1.5, 1.0, 1.2])
    with pytest.raises(IndexError):
        spatial.cutout_mask(lam_lats, lam_lons, g_lats, g_lons, neighbours=4)
    with pytest.raises(IndexError):
        spatial.cutout_mask(lam_lats, lam_lons, g_lats, g_lons, neighbours=4, device=True)
    lam = _regular(10.0, 10.0)
    with pytest.raises(ValueError, match="neighbours <= 16"):
        spatial.cutout_mask(lam["latitudes"][:200] * 0.1 + 45, lam["longitudes"][:200] * 0.05, g_lats, g_lons, neighbours=17, device=True)
    a = spatial.cutout_mask(lam_lats, lam_lons, g_lats, g_lons, neighbours=3, device=True)
    b = spatial.cutout_mask(lam_lats, lam_lons, g_lats, g_lons, neighbours=3)
    assert np.array_equal(a, b) and np.array_equal(b, oracle.cutout_mask(lam_lats, lam_lons, g_lats, g_lons, neighbours=3))
