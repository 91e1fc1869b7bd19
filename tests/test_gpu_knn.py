"""GPU k-NN index build against the reference's cKDTree statement (R: spatial.py:587-635).

Index work is bit-exact: the table returned by ``nearest_grid_points_device`` (default ``ties="ckdtree"``) must EQUAL
cKDTree's, distances included.  The kernel alone (``ties="index"``) may differ from cKDTree only in rows whose
candidates are exactly equidistant, where it orders by source index."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from anemoi_transform_amd import interp, native
from anemoi_transform_amd.grids import lookup
from oracle import oracle

pytestmark = pytest.mark.gpu


def compare(src, tgt, k, max_distance=None):
    args = (src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"])
    kw = dict(max_distance=max_distance, num_neighbours_to_return=k, return_distances=True)
    want_i, want_d = oracle.nearest_grid_points(*args, **kw)
    got_i, got_d = interp.nearest_grid_points_device(*args, **kw)
    assert got_i.shape == want_i.shape and got_i.dtype == np.int64
    assert np.array_equal(got_d, want_d), "distances must be bit-identical to cKDTree's"
    assert np.array_equal(got_i, want_i), "the index table must be cKDTree's"
    # the kernel on its own (ties by source index): same distances; rows differ only where candidates are equidistant
    raw_i, raw_d = interp.nearest_grid_points_device(*args, ties="index", **kw)
    assert np.array_equal(raw_d, want_d)
    want_i2, raw_i2, d2 = want_i.reshape(len(want_i), -1), raw_i.reshape(len(raw_i), -1), raw_d.reshape(len(raw_d), -1)
    sxyz = interp.unit_sphere_xyz(src["latitudes"], src["longitudes"])
    txyz = interp.unit_sphere_xyz(tgt["latitudes"], tgt["longitudes"])
    found = raw_i2 < len(sxyz)
    diff = sxyz[np.where(found, raw_i2, 0)] - txyz[:, None, :]
    d_of_idx = np.sqrt(diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1] + diff[..., 2] * diff[..., 2])
    assert np.array_equal(d_of_idx[found], d2[found])  # every index really lies at its distance: an exact k-NN list
    differ = (want_i2 != raw_i2).any(axis=1)
    return differ.mean()


@pytest.mark.parametrize("k", [1, 2, 4, 7])
def test_knn_matches_ckdtree_o96_to_1deg(dev, k):
    compare(lookup("o96"), lookup([1.0, 1.0]), k)


def test_knn_latlon_source_with_dense_poles(dev):
    """A lat-lon source has hundreds of coincident-ish points at the poles: the tie-heavy case."""
    compare(lookup([2.0, 2.0]), lookup("o48"), 4)


def test_knn_regional_source_and_max_distance(dev):
    rng = np.random.default_rng(3)
    src = dict(latitudes=rng.uniform(40, 60, 5000), longitudes=rng.uniform(0, 30, 5000))
    tgt = lookup([5.0, 5.0])
    compare(src, tgt, 3)
    compare(src, tgt, 2, max_distance=0.05)
    i, d = interp.nearest_grid_points_device(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"],
                                             max_distance=0.05, num_neighbours_to_return=2, return_distances=True)
    assert (i == 5000).any() and np.isinf(d[i == 5000]).all()  # cKDTree's "not found" marker (R: spatial.py:630-632)


def test_knn_tiny_inputs(dev):
    src = dict(latitudes=np.array([0.0, 10.0, -10.0]), longitudes=np.array([0.0, 20.0, 340.0]))
    assert np.array_equal(interp.nearest_grid_points_device(src["latitudes"], src["longitudes"], src["latitudes"], src["longitudes"]),
                          [0, 1, 2])
    i = interp.nearest_grid_points_device(src["latitudes"], src["longitudes"], np.array([9.0]), np.array([19.0]),
                                          num_neighbours_to_return=3)
    assert i.tolist() == [[1, 0, 2]]
    one = dict(latitudes=np.array([5.0]), longitudes=np.array([5.0]))
    assert interp.nearest_grid_points_device(one["latitudes"], one["longitudes"], src["latitudes"], src["longitudes"]).tolist() == [0, 0, 0]


def test_knn_full_size_o1280_to_quarter_degree(dev):
    """BASELINE size: the index tables of the headline benchmark (k = 4) and of method="nearest" (k = 1), GPU vs cKDTree:
    identical, equidistant candidates included."""
    import time

    src, tgt = lookup("o1280"), lookup("0.25")
    args = (src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"])
    for k in (4, 1):
        t0 = time.perf_counter()
        want_i, want_d = oracle.nearest_grid_points(*args, num_neighbours_to_return=k, return_distances=True)
        t_cpu = time.perf_counter() - t0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        raw_i, raw_d = interp.nearest_grid_points_device(*args, num_neighbours_to_return=k, return_distances=True, ties="index")
        t_raw = time.perf_counter() - t0
        t0 = time.perf_counter()
        got_i, got_d = interp.nearest_grid_points_device(*args, num_neighbours_to_return=k, return_distances=True)
        t_gpu = time.perf_counter() - t0
        assert np.array_equal(got_d, want_d) and np.array_equal(raw_d, want_d)
        assert np.array_equal(got_i, want_i)
        differ = (raw_i.reshape(len(raw_i), -1) != want_i.reshape(len(want_i), -1)).any(axis=1)
        print(f"\nO1280->0.25 k={k}: cKDTree {t_cpu:.2f} s; device, ties by index {t_raw:.2f} s ({differ.mean():.4%} rows in another "
              f"order); device, ties settled by cKDTree {t_gpu:.2f} s (identical table)")


def test_device_mask_builders_equal_the_host_ones(dev):
    """cutout_mask / global_on_lam_mask with device=True (atx_knn + atx_cutout_inside) against the host builders
    (which are pinned to the reference's known answers in tests/test_spatial.py) — and on those known answers."""
    from anemoi_transform_amd import spatial

    def lam_grid(lat0, lat1, lon0, lon1, n):
        lats, lons = np.meshgrid(np.linspace(lat0, lat1, n), np.linspace(lon0, lon1, n))
        return lats.flatten(), lons.flatten()

    lam_lats, lam_lons = lam_grid(44.0, 46.0, 0.0, 2.0, 11)
    g_lats = np.array([43.1, 44.0, 45.0, 45.5, 46.0, 50.0])
    g_lons = np.array([359.1, 359.5, 0.0, 1.0, 2.0, 0.0])
    for cd in (1.0, 3.0, 5.0):  # R: tests/test_spatial.py:20-50
        m = spatial.cutout_mask(lam_lats, lam_lons, g_lats, g_lons, cropping_distance=cd, max_distance_km=250.0, device=True)
        assert np.array_equal(m, [True, False, False, False, False, False])
    m = spatial.cutout_mask(lam_lats, lam_lons, np.array([44.0, 45.0, 46.0, 46.1, 47.5]), np.array([0.0, 1.0, 2.0, -0.1, -1.5]),
                            min_distance_km=100.0, device=True)
    assert np.array_equal(m, [False, False, False, False, True])  # R: tests/test_spatial.py:53-79

    rng = np.random.default_rng(12)
    lam_lats, lam_lons = lam_grid(35.0, 55.0, -5.0, 20.0, 60)
    lam_lats = lam_lats + rng.normal(0, 0.02, lam_lats.shape)
    glob = lookup("o160")
    for kw in (dict(), dict(min_distance_km=50.0), dict(max_distance_km=500.0, neighbours=4)):
        a = spatial.cutout_mask(lam_lats, lam_lons, glob["latitudes"], glob["longitudes"], device=True, **kw)
        b = spatial.cutout_mask(lam_lats, lam_lons, glob["latitudes"], glob["longitudes"], **kw)
        assert np.array_equal(a, b), kw
        assert a.any() and not a.all()
    a = spatial.global_on_lam_mask(lam_lats, lam_lons, glob["latitudes"], glob["longitudes"], distance_km=40.0, device=True)
    b = spatial.global_on_lam_mask(lam_lats, lam_lons, glob["latitudes"], glob["longitudes"], distance_km=40.0)
    assert np.array_equal(a, b) and len(a) > 100
    # thinning_mask (R: spatial.py:443-503): nearest LAM point of every global point of the surrounding box.
    # Longitudes in one convention (the box test of cropping_mask does not wrap); the jittered LAM latitudes make ties unlikely
    glob_lons = np.where(glob["longitudes"] > 180.0, glob["longitudes"] - 360.0, glob["longitudes"])
    a = spatial.thinning_mask(lam_lats, lam_lons, glob["latitudes"], glob_lons, device=True)
    b = spatial.thinning_mask(lam_lats, lam_lons, glob["latitudes"], glob_lons)
    assert a.shape == b.shape and len(a) > 100
    assert np.array_equal(a, b)
    # the same on an UNjittered LAM grid inside a regular global grid: equidistant candidates everywhere
    lam_lats, lam_lons = lam_grid(40.0, 50.0, 0.0, 10.0, 21)
    reg = lookup([1.0, 1.0])
    reg_lons = np.where(reg["longitudes"] > 180.0, reg["longitudes"] - 360.0, reg["longitudes"])
    a = spatial.thinning_mask(lam_lats, lam_lons, reg["latitudes"], reg_lons, device=True)
    assert np.array_equal(a, spatial.thinning_mask(lam_lats, lam_lons, reg["latitudes"], reg_lons)) and len(a) > 100
    for kw in (dict(), dict(neighbours=4, min_distance_km=30.0)):
        a = spatial.cutout_mask(lam_lats, lam_lons, reg["latitudes"], reg_lons, device=True, **kw)
        assert np.array_equal(a, spatial.cutout_mask(lam_lats, lam_lons, reg["latitudes"], reg_lons, **kw)), kw
    with pytest.raises(IndexError):  # more neighbours than LAM points fails like the host path, before any launch
        spatial.cutout_mask(lam_lats[:3], lam_lons[:3], reg["latitudes"], reg_lons, neighbours=4, device=True)


@pytest.mark.parametrize("seed", range(8))
def test_knn_random_point_sets(dev, seed):
    """Seeded random cases: tiny and odd-sized source sets (around the 8-point leaves of the box tree), clustered points,
    exact duplicates, more neighbours asked for than there are sources (cKDTree pads with index n / distance inf)."""
    rng = np.random.default_rng(500 + seed)
    for case in range(6):
        n_src = int(rng.choice([1, 2, 7, 8, 9, 15, 17, 63, 65, 500, 3001]))
        n_tgt = int(rng.choice([1, 5, 64, 257, 1500]))
        k = int(rng.choice([1, 2, 3, 4, 8]))
        style = rng.choice(["uniform", "cluster", "duplicates"])
        if style == "uniform":
            lat, lon = np.degrees(np.arcsin(rng.uniform(-1, 1, n_src))), rng.uniform(0, 360, n_src)
        elif style == "cluster":
            lat, lon = 45 + rng.normal(0, 0.01, n_src), 10 + rng.normal(0, 0.01, n_src)
        else:
            base = max(1, n_src // 3)
            pick = rng.integers(0, base, n_src)
            lat, lon = np.degrees(np.arcsin(rng.uniform(-1, 1, base)))[pick], rng.uniform(0, 360, base)[pick]
        src = dict(latitudes=lat, longitudes=lon)
        tlat, tlon = np.degrees(np.arcsin(rng.uniform(-1, 1, n_tgt))), rng.uniform(0, 360, n_tgt)
        if style == "cluster":
            tlat, tlon = 45 + rng.normal(0, 0.02, n_tgt), 10 + rng.normal(0, 0.02, n_tgt)
        tgt = dict(latitudes=tlat, longitudes=tlon)
        want_i, want_d = oracle.nearest_grid_points(lat, lon, tlat, tlon, num_neighbours_to_return=k, return_distances=True)
        got_i, got_d = interp.nearest_grid_points_device(lat, lon, tlat, tlon, num_neighbours_to_return=k, return_distances=True)
        what = f"seed {seed} case {case}: {style} n_src={n_src} n_tgt={n_tgt} k={k}"
        assert np.array_equal(got_i, want_i), what + ": the index table must be cKDTree's (duplicates and ties included)"
        got_i, got_d = interp.nearest_grid_points_device(lat, lon, tlat, tlon, num_neighbours_to_return=k, return_distances=True,
                                                         ties="index")  # below: the kernel on its own
        assert got_i.shape == want_i.shape, what
        assert np.array_equal(got_d, want_d), what + ": distances must be bit-identical to cKDTree's"
        missing = np.isinf(want_d)
        assert np.array_equal(got_i[missing], want_i[missing]), what + ": the 'no neighbour' marker is len(source)"
        # where all candidate distances of a row are distinct the index lists are identical
        gi, wi, gd = got_i.reshape(n_tgt, -1), want_i.reshape(n_tgt, -1), got_d.reshape(n_tgt, -1)
        with np.errstate(invalid="ignore"):  # inf - inf between two missing neighbours
            distinct = (np.diff(gd, axis=1) != 0).all(axis=1) if gd.shape[1] > 1 else np.ones(n_tgt, bool)
        if style != "duplicates":  # duplicated sources tie at any rank: only the distances are comparable
            assert np.array_equal(gi[distinct], wi[distinct]), what
