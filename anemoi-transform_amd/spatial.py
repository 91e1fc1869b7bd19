"""Mask / index builders whose OUTPUTS feed the gather kernels (CPU precompute, one-off).

Same call surface as the reference's ``anemoi.transform.spatial`` builders
(R: spatial.py:236-275 ``cropping_mask``, :294-440 ``cutout_mask``, :443-503
``thinning_mask``, :506-536 ``global_on_lam_mask``): they return boolean masks or
index lists that ``regrid(mask=...)`` (R: filters/fields/regrid.py:384-429) or a
``GatherPlan`` then applies on the GPU.  SURVEY.md §8 row a8: these run once per
grid pair and stay on the host; the reference's per-point Python loop over
Möller–Trumbore ray/triangle tests (R: spatial.py:404-424) is evaluated here for
all points at once with numpy (same arithmetic per triangle, same ``any``-of-
neighbours rule).

Constants: the reference imports ``R_earth`` and ``radian`` from earthkit-meteo
(R: constants.py:11-14), which is not installed; ``R_earth = 6 371 229 m`` and
``radian = pi / 180`` are used (the latter is what R: spatial.py:364's comment
states; the former is earthkit-meteo's value as recalled in SURVEY.md §8c — the
reference's known answers, tests/test_spatial.py:25-79, hold for it).
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .interp import unit_sphere_xyz

R_earth_km = 6371.229
radian = np.pi / 180.0


def cropping_mask(lats, lons, north: float, west: float, south: float, east: float) -> np.ndarray:
    """Points inside a lat/lon box, longitudes matched modulo 360 (R: spatial.py:236-275)."""
    return (
        (lats >= south)
        & (lats <= north)
        & (
            ((lons >= west) & (lons <= east))
            | ((lons >= west + 360) & (lons <= east + 360))
            | ((lons >= west - 360) & (lons <= east - 360))
        )
    )


def _check_latlon_arrays(lats, lons, global_lats, global_lons) -> None:
    assert global_lats.ndim == 1 and global_lons.ndim == 1 and lats.ndim == 1 and lons.ndim == 1
    assert global_lats.shape == global_lons.shape
    assert lats.shape == lons.shape


def _resolution(points: np.ndarray) -> float:
    from scipy.spatial import cKDTree

    distances, _ = cKDTree(points).query(points, k=2)
    return np.min(distances[:, 1])


def _distance_km_to_resolution(distance_km: Any, lam_points, global_points) -> float:
    # R: spatial.py:101-108
    if isinstance(distance_km, (int, float)):
        return distance_km / R_earth_km
    return _resolution({"lam": lam_points, "global": global_points, None: global_points}[distance_km])


def _row_dots(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """``np.dot(a[i], b[i])`` for every row, with np.dot's OWN bits: a batched ``matmul`` of ``[1, 3] @ [3, 1]`` goes through the
    same ``cblas_ddot`` as the reference's per-point ``np.dot`` (R: spatial.py:213,218,222,225) — on x86-64 OpenBLAS that is a
    chain of fused multiply-adds, which ``einsum`` / ``(a * b).sum(1)`` do not reproduce.  A point lying exactly on a triangle
    edge (regular LAM grid, global point on one of its meridians) is decided by that last bit."""
    return np.matmul(a[:, None, :], b[:, :, None])[:, 0, 0]


def rays_hit_triangles(directions: np.ndarray, v0: np.ndarray, v1: np.ndarray, v2: np.ndarray) -> np.ndarray:
    """Möller–Trumbore for rays from the origin: ``directions [N, 3]`` against triangles ``[N, 3]`` each
    (R: spatial.py:186-233, evaluated for all N at once with the per-point arithmetic of np.cross / np.dot)."""
    epsilon = 0.0000001
    directions = np.ascontiguousarray(directions, dtype=np.float64)
    e1, e2 = v1 - v0, v2 - v0
    h = np.cross(directions, e2)
    a = _row_dots(e1, h)
    ok = ~((-epsilon < a) & (a < epsilon))
    with np.errstate(divide="ignore", invalid="ignore"):
        f = 1.0 / a
        s = 0.0 - v0
        u = f * _row_dots(s, h)
        ok &= ~((u < 0.0) | (u > 1.0))
        q = np.cross(s, e1)
        v = f * _row_dots(directions, q)
        ok &= ~((v < 0.0) | (u + v > 1.0))
        t = f * _row_dots(e2, q)
    return ok & (t > epsilon)


def cutout_mask(
    lats,
    lons,
    global_lats,
    global_lons,
    cropping_distance: float = 2.0,
    neighbours: int = 5,
    min_distance_km: int | float | None = None,
    max_distance_km: int | float | None = None,
    plot: str | None = None,
    device: bool = False,
) -> np.ndarray:
    """Mask of the GLOBAL points to keep around a limited-area grid: ``False`` for points inside the
    LAM, closer than ``min_distance_km`` or further than ``max_distance_km`` (R: spatial.py:294-440).

    ``device=True`` runs the neighbour search (``atx_knn_*``) and the ray/triangle tests
    (``atx_cutout_inside``) on the GPU — SURVEY.md §8f rank 4; same result."""
    assert cropping_distance >= 0.0, "cropping_distance must be non-negative"
    assert min_distance_km is None or min_distance_km >= 0.0, "min_distance_km must be non-negative"
    assert max_distance_km is None or max_distance_km >= 0.0, "max_distance_km must be non-negative"
    assert neighbours > 0, "neighbours must be positive"
    from scipy.spatial import cKDTree

    _check_latlon_arrays(lats, lons, global_lats, global_lons)
    north, south, east, west = np.amax(lats), np.amin(lats), np.amax(lons), np.amin(lons)

    effective = cropping_distance
    if max_distance_km is not None:
        max_lat = max(abs(north), abs(south))
        one_degree_km = R_earth_km * np.cos(np.deg2rad(max_lat)) * radian
        effective = max(cropping_distance, 1.1 * max_distance_km / one_degree_km)

    mask = cropping_mask(
        global_lats, global_lons, np.min([90.0, north + effective]), west - effective,
        np.max([-90.0, south - effective]), east + effective,
    )
    global_points = unit_sphere_xyz(global_lats[mask], global_lons[mask])
    lam_points = unit_sphere_xyz(lats, lons)
    min_distance = _distance_km_to_resolution(min_distance_km, lam_points, global_points)

    if device:
        distances, inside = _device_neighbours_and_inside(lam_points, global_points, neighbours)
    else:
        distances, indices = cKDTree(lam_points).query(global_points, k=neighbours)
        distances = distances.reshape(len(global_points), -1)
        indices = indices.reshape(len(global_points), -1)
        inside = np.zeros(len(global_points), dtype=bool)
        for j in range(neighbours):  # any of the `neighbours` triangles of consecutive nearest points
            inside |= rays_hit_triangles(
                global_points,
                lam_points[indices[:, j]],
                lam_points[indices[:, (j + 1) % neighbours]],
                lam_points[indices[:, (j + 2) % neighbours]],
            )
    nearest = np.min(distances, axis=1) if len(global_points) else np.zeros(0)
    exclude = inside | (nearest <= min_distance)
    if max_distance_km is not None:
        exclude |= nearest > (max_distance_km / R_earth_km)

    too_far_mask: Any = False
    if isinstance(max_distance_km, (int, float)):
        too_far_mask = ~mask.copy()  # everything outside the cropping box is too far
    mask[mask] = exclude
    mask[too_far_mask] = True
    return ~mask


def _device_neighbours_and_inside(lam_points: np.ndarray, global_points: np.ndarray, neighbours: int):
    """k nearest LAM points of every global point (``interp.device_knn``: cKDTree's own order among equidistant
    points, since the triangles are formed from CONSECUTIVE neighbours) and the ray/triangle verdict, on the GPU."""
    import torch

    from . import native
    from . import stack as _stack
    from .interp import MAX_DEVICE_K, device_knn

    if neighbours > len(lam_points):
        # the host path fails the same way: cKDTree pads with the index len(lam_points), which then indexes out of range
        raise IndexError(f"index {len(lam_points)} is out of bounds for axis 0 with size {len(lam_points)} "
                         f"(neighbours={neighbours} > {len(lam_points)} limited-area points)")
    if neighbours > MAX_DEVICE_K:
        raise ValueError(f"cutout_mask(device=True) supports neighbours <= {MAX_DEVICE_K}, got {neighbours}; use device=False")
    if len(global_points) == 0:
        return np.zeros((0, neighbours)), np.zeros(0, dtype=bool)
    dev = _stack.device()
    idx, d2, _ = device_knn(lam_points, global_points, neighbours)
    lam_d = torch.from_numpy(np.ascontiguousarray(lam_points)).to(dev)
    glob_d = torch.from_numpy(np.ascontiguousarray(global_points)).to(dev)
    inside = native.cutout_inside(glob_d, lam_d, torch.from_numpy(idx.astype(np.int32)).to(dev))
    return np.sqrt(d2), inside.cpu().numpy().astype(bool)


def thinning_mask(lats, lons, global_lats, global_lons, cropping_distance: float = 2.0, device: bool = False) -> np.ndarray:
    """Indices of the LAM points closest to each global point of the surrounding box (R: spatial.py:443-503).
    ``device=True``: the k = 1 search runs on the GPU (``atx_knn_*``; equidistant candidates settled by cKDTree: same indices)."""
    from scipy.spatial import cKDTree

    _check_latlon_arrays(lats, lons, global_lats, global_lons)
    north, south, east, west = np.amax(lats), np.amin(lats), np.amax(lons), np.amin(lons)
    mask = cropping_mask(
        global_lats, global_lons, np.min([90.0, north + cropping_distance]), west - cropping_distance,
        np.max([-90.0, south - cropping_distance]), east + cropping_distance,
    )
    global_points = unit_sphere_xyz(global_lats[mask], global_lons[mask])
    if device:
        from .interp import nearest_grid_points_device

        return nearest_grid_points_device(lats, lons, global_lats[mask], global_lons[mask])
    _, indices = cKDTree(unit_sphere_xyz(lats, lons)).query(global_points, k=1)
    return indices


def global_on_lam_mask(lats, lons, global_lats, global_lons, distance_km: float | None = None, device: bool = False) -> np.ndarray:
    """Sorted unique indices of the global points within ``distance_km`` of any LAM point
    (R: spatial.py:506-536) — the ``mask`` of ``regrid(mask=...)``.  ``device=True``: one k = 1
    search of the LAM points from every global point on the GPU instead of a ball query per LAM point."""
    from scipy.spatial import cKDTree

    _check_latlon_arrays(lats, lons, global_lats, global_lons)
    global_points = unit_sphere_xyz(global_lats, global_lons)
    lam_points = unit_sphere_xyz(lats, lons)
    distance = _distance_km_to_resolution(distance_km, lam_points, global_points)
    if device:
        import torch

        from . import native
        from . import stack as _stack

        dev = _stack.device()
        lam_d = torch.from_numpy(np.ascontiguousarray(lam_points)).to(dev)
        glob_d = torch.from_numpy(np.ascontiguousarray(global_points)).to(dev)
        _, d2 = native.KnnIndex(lam_d).query(glob_d, 1)
        return np.flatnonzero(d2.cpu().numpy()[:, 0] <= distance * distance)
    found = cKDTree(global_points).query_ball_point(lam_points, distance)
    return np.array(sorted(set(i for sub in found for i in sub)))
