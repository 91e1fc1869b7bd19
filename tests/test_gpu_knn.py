"""GPU k-NN index build against the reference's cKDTree statement (R: spatial.py:587-635).

Squared distances must be bit-identical to scipy's; index lists must agree wherever the candidate
distances are distinct (exact ties are ordered by index here, by traversal in cKDTree)."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from anemoi_transform_amd import interp, native
from anemoi_transform_amd.grids import lookup
from oracle import oracle

pytestmark = pytest.mark.gpu


def compare(src, tgt, k, max_distance=None):
    want_i, want_d = oracle.nearest_grid_points(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"],
                                                max_distance=max_distance, num_neighbours_to_return=k, return_distances=True)
    got_i, got_d = interp.nearest_grid_points_device(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"],
                                                     max_distance=max_distance, num_neighbours_to_return=k, return_distances=True)
    assert got_i.shape == want_i.shape and got_i.dtype == np.int64
    assert np.array_equal(got_d, want_d), "distances must be bit-identical to cKDTree's"
    want_i2, got_i2 = want_i.reshape(len(want_i), -1), got_i.reshape(len(got_i), -1)
    got_d2 = got_d.reshape(len(got_d), -1)
    # every returned index really lies at the returned distance (same float64 arithmetic as scipy):
    # together with "distances identical to cKDTree's" this makes the answer an exact k-NN list
    sxyz = interp.unit_sphere_xyz(src["latitudes"], src["longitudes"])
    txyz = interp.unit_sphere_xyz(tgt["latitudes"], tgt["longitudes"])
    found = got_i2 < len(sxyz)
    diff = sxyz[np.where(found, got_i2, 0)] - txyz[:, None, :]
    d_of_idx = np.sqrt(diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1] + diff[..., 2] * diff[..., 2])
    assert np.array_equal(d_of_idx[found], got_d2[found])
    # rows may differ from cKDTree only where two candidates are EXACTLY equidistant
    differ = (want_i2 != got_i2).any(axis=1)
    if differ.any() and got_i2.shape[1] > 1:
        rows = np.flatnonzero(differ)
        tied = (np.diff(got_d2[rows], axis=1) == 0).any(axis=1)
        k_th_tie = ~tied  # a tie between the k-th and the (k+1)-th candidate: same distances, other point
        assert np.array_equal(np.sort(d_of_idx[rows], axis=1), got_d2[rows])
        assert (tied | k_th_tie).all()
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
    """BASELINE size: the k=4 index table of the headline benchmark, GPU vs cKDTree."""
    import time

    src, tgt = lookup("o1280"), lookup("0.25")
    t0 = time.perf_counter()
    want_i, want_d = oracle.nearest_grid_points(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"],
                                                num_neighbours_to_return=4, return_distances=True)
    t_cpu = time.perf_counter() - t0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got_i, got_d = interp.nearest_grid_points_device(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"],
                                                     num_neighbours_to_return=4, return_distances=True)
    t_gpu = time.perf_counter() - t0
    assert np.array_equal(got_d, want_d)
    differ = (got_i != want_i).any(axis=1)
    print(f"\nO1280->0.25 k=4: cKDTree {t_cpu:.2f} s, device (incl. host xyz + copies) {t_gpu:.2f} s, rows differing by ties {differ.mean():.4%}")
    same = ~differ
    assert np.array_equal(got_i[same], want_i[same])
    assert differ.mean() < 0.05


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
    same = a == b
    xyz = spatial.unit_sphere_xyz(lam_lats, lam_lons)
    assert same.mean() > 0.99  # any difference is an exact tie ...
    box = spatial.cropping_mask(glob["latitudes"], glob_lons, lam_lats.max() + 2.0, lam_lons.min() - 2.0, lam_lats.min() - 2.0, lam_lons.max() + 2.0)
    g = spatial.unit_sphere_xyz(glob["latitudes"][box], glob_lons[box])
    da, db = ((g - xyz[a]) ** 2).sum(axis=1), ((g - xyz[b]) ** 2).sum(axis=1)
    assert np.array_equal(da, db)  # ... i.e. the chosen neighbours are equally near


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
