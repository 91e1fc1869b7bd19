"""CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy / scipy restatement of the filter hot path of ecmwf/anemoi-transform
0.4.2 (regrid gather / sparse interpolation, per-point transforms, mask
filters).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import this module, and only as the checker.  The
product package never imports it: its compute path is libatx (HIP) and fails
loudly without it.

Every function cites the reference statement it restates (``R:`` = paths under
``/root/reference/src/anemoi/transform/``).  numpy and scipy are the libraries
that execute those statements in the reference, so the restatement runs the
identical native routines (``take``, ufuncs, ``csr_matvec``, ``cKDTree``).

Pinning: the reference package cannot be imported in
the build container (``ModuleNotFoundError: earthkit`` / ``anemoi.utils`` —
ordinary missing dependencies, no network to install them), so the oracle is
pinned by the reference's own test vectors, transcribed as data into
``tests/golden/reference_vectors.json`` and checked in
``tests/test_oracle_golden.py``.  Regrid *numerics* are pinned by nothing in the
reference (``tests/field_filters/test_regrid.py`` only iterates the pipeline);
for K1-K3 the oracle is the reference's statement itself on synthetic inputs.

Fields are plain dicts here: ``{"param": str, "values": ndarray, "latitudes":
1-D ndarray (one per grid point), "longitudes": ..., **metadata}``.
"""

from __future__ import annotations

from typing import Any

import numpy as np

# R: constants.py:13  g from earthkit.meteo; value pinned by
# R: filters/tabular/geopotential_to_height.py:51
G = 9.80665

# R: filters/fields/apply_mask.py:23-36
OPERATORS = {
    ">": np.greater,
    "<": np.less,
    "==": np.equal,
    "!=": np.not_equal,
    ">=": np.greater_equal,
    "<=": np.less_equal,
    "gt": np.greater,
    "lt": np.less,
    "eq": np.equal,
    "ne": np.not_equal,
    "ge": np.greater_equal,
    "le": np.less_equal,
}


# --------------------------------------------------------------------------------
# array-level statements
# --------------------------------------------------------------------------------
def gather_nn(data: np.ndarray, nearest_grid_points: np.ndarray) -> np.ndarray:
    """R: filters/fields/regrid.py:380 ``data = data[..., self.nearest_grid_points]``."""
    return data[..., nearest_grid_points]


def csr_apply(matrix_data, matrix_indices, matrix_indptr, matrix_shape, data: np.ndarray) -> np.ndarray:
    """R: regrid.py:283-285,310 ``csr_array((data, indices, indptr), shape) @ data``."""
    from scipy.sparse import csr_array

    matrix = csr_array((matrix_data, matrix_indices, matrix_indptr), shape=tuple(int(s) for s in matrix_shape))
    return matrix @ data


def masked_subset(data: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """R: regrid.py:420 ``data = data[..., self.mask]`` (bool or integer mask)."""
    return data[..., mask]


def compute_mask(mask_values: np.ndarray, *, mask_value=None, threshold=None, threshold_operator=">") -> np.ndarray:
    """R: apply_mask.py:160-163."""
    if threshold is not None:
        return OPERATORS[threshold_operator](mask_values, threshold)
    return mask_values == mask_value


def apply_mask_values(values: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """R: apply_mask.py:184-185 ``values[self.mask] = np.nan`` on the flattened copy."""
    values = np.array(values, dtype=values.dtype, copy=True).reshape(-1)
    values[mask] = np.nan
    return values


def not_nan_mask(data: np.ndarray) -> np.ndarray:
    """R: filters/fields/remove_nans.py:101 ``~np.isnan(data)``."""
    return ~np.isnan(data)


def rescale_forward(x, scale, offset):
    """R: filters/fields/rescale.py:25 ``x * self.scale + self.offset``."""
    return x * scale + offset


def rescale_backward(x, scale, offset):
    """R: rescale.py:28 ``(x - self.offset) / self.scale``."""
    return (x - offset) / scale


def orog_to_z(x):
    """R: filters/fields/orog_to_z.py:59 ``orography.to_numpy() * g``."""
    return x * G


def z_to_orog(x):
    """R: orog_to_z.py:77 ``geopotential.to_numpy() / g``."""
    return x / G


def clip(x, minimum, maximum):
    """R: filters/fields/clipper.py:69 ``np.clip(data, self.minimum, self.maximum)``."""
    return np.clip(x, minimum, maximum)


def impute_nans(x, value):
    """R: filters/fields/impute_nans.py:53-54."""
    values = np.array(x, copy=True).reshape(-1)
    values[np.isnan(values)] = value
    return values


def lnsp_to_sp(x):
    """R: filters/fields/lnsp_to_sp.py:47 ``np.exp(...)``."""
    return np.exp(x)


def sp_to_lnsp(x):
    """R: lnsp_to_sp.py:65 ``np.log(...)``."""
    return np.log(x)


# --------------------------------------------------------------------------------
# index / geometry precompute
# --------------------------------------------------------------------------------
def latlon_to_xyz(lat, lon, radius: float = 1.0):
    """R: spatial.py:132-167 (unit sphere, h = 0)."""
    phi = np.deg2rad(lat)
    lda = np.deg2rad(lon)
    cos_phi = np.cos(phi)
    return cos_phi * np.cos(lda) * radius, cos_phi * np.sin(lda) * radius, np.sin(phi) * radius


def nearest_grid_points(
    source_latitudes,
    source_longitudes,
    target_latitudes,
    target_longitudes,
    max_distance=None,
    num_neighbours_to_return: int = 1,
    return_distances: bool = False,
):
    """R: spatial.py:587-635 — cKDTree on unit-sphere xyz, chord distance; returns
    indices (int64) or ``(indices, distances)`` in THAT order (spatial.py:633-635)."""
    from scipy.spatial import cKDTree

    source_points = np.array(latlon_to_xyz(source_latitudes, source_longitudes)).transpose()
    target_points = np.array(latlon_to_xyz(target_latitudes, target_longitudes)).transpose()
    if max_distance is None:
        distances, indices = cKDTree(source_points).query(target_points, k=num_neighbours_to_return)
    else:
        distances, indices = cKDTree(source_points).query(
            target_points, k=num_neighbours_to_return, distance_upper_bound=max_distance
        )
    if return_distances:
        return indices, distances
    return indices


# --------------------------------------------------------------------------------
# filter-level restatements on dict fields
# --------------------------------------------------------------------------------
def _flat(field: dict) -> np.ndarray:
    # R: fields.py:178-202 to_numpy(flatten=True) returns a flattened copy
    return np.asarray(field["values"]).flatten()


def _selected(field: dict, param) -> bool:
    # R: fields.py:767-797 FieldSelection(param=...).match
    if param is None:
        return True
    params = (param,) if isinstance(param, str) else tuple(param)
    if len(params) == 0:
        return True
    return field.get("param") in params


def filter_apply_mask(
    fields: list[dict],
    *,
    mask_values: np.ndarray | None = None,
    mask_param: str | None = None,
    mask_value=None,
    threshold=None,
    threshold_operator: str = ">",
    rename: str | None = None,
    param=None,
    return_mask: bool = False,
) -> list[dict]:
    """R: apply_mask.py:114-245 (``mask_values`` stands for the flattened first field of ``path``)."""
    if (mask_values is None) == (mask_param is None):
        raise ValueError("Exactly one of `path` or `mask_param` must be provided.")
    if (mask_value is None) == (threshold is None):
        raise ValueError("Exactly one of `mask_value` or `threshold` must be provided.")
    if threshold is not None and threshold_operator not in OPERATORS:
        raise ValueError(f"Invalid threshold operator: {threshold_operator}.")
    kw = dict(mask_value=mask_value, threshold=threshold, threshold_operator=threshold_operator)

    if mask_param is None:
        mask = compute_mask(np.asarray(mask_values).flatten(), **kw)
        remaining = list(fields)
    else:
        mask_field = None
        remaining = []
        for f in fields:  # R: apply_mask.py:194-218
            if f.get("param") == mask_param:
                if mask_field is None:
                    mask_field = f
                if not return_mask:
                    continue
            remaining.append(f)
        if mask_field is None:
            raise ValueError(f"Mask parameter '{mask_param}' not found in input data.")
        mask = compute_mask(_flat(mask_field), **kw)

    out = []
    for f in remaining:
        if _selected(f, param):
            g = dict(f)
            g["values"] = apply_mask_values(_flat(f), mask)
            if rename is not None:
                g["param"] = f"{f['param']}_{rename}"
            out.append(g)
        else:
            out.append(f)
    return out


def filter_remove_nans(fields: list[dict], *, param: str | None = None) -> list[dict]:
    """R: remove_nans.py:75-119 — mask from the first field (or first with ``param``)."""
    if param is None:
        first = fields[0]
    else:
        for first in fields:
            if first.get("param") == param:
                break
        else:
            raise ValueError(f"{param=} not found")
    mask = not_nan_mask(_flat(first))
    lat = np.asarray(first["latitudes"])[mask]
    lon = np.asarray(first["longitudes"])[mask]
    out = []
    for f in fields:
        g = dict(f)
        g["values"] = _flat(f)[mask]
        g["latitudes"], g["longitudes"] = lat, lon
        out.append(g)
    return out


def _map_selected(fields, selected_param, fn, **new_metadata):
    # R: filter.py:188-196: unselected fields pass through by identity
    out = []
    for f in fields:
        if _selected(f, selected_param):
            g = dict(f)
            g["values"] = fn(np.asarray(f["values"]))
            g.update({k: v for k, v in new_metadata.items()})
            out.append(g)
        else:
            out.append(f)
    return out


def filter_rescale(fields, *, scale, offset, param, backward: bool = False):
    """R: rescale.py:31-66."""
    if backward:
        return _map_selected(fields, param, lambda x: rescale_backward(x, scale, offset), param=param)
    return _map_selected(fields, param, lambda x: rescale_forward(x, scale, offset), param=param)


def filter_orog_to_z(fields, *, orography="orog", geopotential="z", backward: bool = False):
    """R: orog_to_z.py:19-78."""
    if backward:
        return _map_selected(fields, geopotential, z_to_orog, param=orography)
    return _map_selected(fields, orography, orog_to_z, param=geopotential)


def filter_clip(fields, *, param, minimum=None, maximum=None):
    """R: clipper.py:58-70."""
    if minimum is None and maximum is None:
        raise ValueError("At least one value for minimum or maximum must be specified.")
    return _map_selected(fields, param, lambda x: clip(x, minimum, maximum), param=param)


def filter_impute_nans(fields, *, param, value):
    """R: impute_nans.py:47-55."""
    return _map_selected(fields, param, lambda x: impute_nans(x, value))


def filter_lnsp_to_sp(fields, *, log_of_surface_pressure="lnsp", surface_pressure="sp", backward: bool = False):
    """R: lnsp_to_sp.py:19-66."""
    if backward:
        return _map_selected(fields, surface_pressure, sp_to_lnsp, param=log_of_surface_pressure)
    return _map_selected(fields, log_of_surface_pressure, lnsp_to_sp, param=surface_pressure)


def interpolator_name(*, method=None, matrix=None, mask=None) -> str:
    """R: regrid.py:432-467 dispatch order: matrix > mask > method == 'nearest' > earthkit."""
    if matrix is not None:
        return "MIRMatrix"
    if mask is not None:
        return "MaskedRegrid"
    if method == "nearest":
        return "ScipyKDTreeNearestNeighbours"
    return "EarthkitRegrid"


def filter_regrid_nearest(fields, *, in_grid: dict | None, out_grid: dict) -> list[dict]:
    """R: regrid.py:315-381 — k = 1 cKDTree gather; index computed once from the first field."""
    idx = None
    out = []
    for f in fields:
        if in_grid is None:
            in_grid = dict(latitudes=np.asarray(f["latitudes"]), longitudes=np.asarray(f["longitudes"]))
        if idx is None:
            idx = nearest_grid_points(
                in_grid["latitudes"], in_grid["longitudes"], out_grid["latitudes"], out_grid["longitudes"]
            )
        data = _flat(f)
        assert data.shape == in_grid["latitudes"].shape, (data.shape, in_grid["latitudes"].shape)
        assert data.shape == in_grid["longitudes"].shape, (data.shape, in_grid["longitudes"].shape)
        g = dict(f)
        g["values"] = gather_nn(data, idx)
        g["latitudes"], g["longitudes"] = out_grid["latitudes"], out_grid["longitudes"]
        out.append(g)
    return out


def filter_regrid_matrix(fields, *, matrix: dict[str, Any]) -> list[dict]:
    """R: regrid.py:262-312 — ``matrix`` is the loaded npz dict (regrid.py:281)."""
    out = []
    for f in fields:
        g = dict(f)
        g["values"] = csr_apply(
            matrix["matrix_data"], matrix["matrix_indices"], matrix["matrix_indptr"], matrix["matrix_shape"], _flat(f)
        )
        g["latitudes"], g["longitudes"] = matrix["out_latitudes"], matrix["out_longitudes"]
        out.append(g)
    return out


def filter_regrid_mask(fields, *, mask: np.ndarray) -> list[dict]:
    """R: regrid.py:384-429 — subset by an index / boolean mask; lat/lon from the first field."""
    out_lat = out_lon = None
    out = []
    for f in fields:
        g = dict(f)
        g["values"] = masked_subset(_flat(f), mask)
        if out_lat is None:
            out_lat = np.asarray(f["latitudes"])[mask]
            out_lon = np.asarray(f["longitudes"])[mask]
        g["latitudes"], g["longitudes"] = out_lat, out_lon
        out.append(g)
    return out


# --------------------------------------------------------------------------------
# mask / index builders (R: spatial.py) — per-point loops kept as in the reference
# --------------------------------------------------------------------------------
# R: constants.py:11-25 — earthkit-meteo constants (package absent): R_earth = 6371229 m as recalled
# in SURVEY.md §8c, radian = pi/180 as stated by the comment at R: spatial.py:364.
R_EARTH_KM = 6371229.0 / 1000
RADIAN = np.pi / 180.0


def cropping_mask(lats, lons, north, west, south, east):
    """R: spatial.py:236-275."""
    return (
        (lats >= south)
        & (lats <= north)
        & (
            ((lons >= west) & (lons <= east))
            | ((lons >= west + 360) & (lons <= east + 360))
            | ((lons >= west - 360) & (lons <= east - 360))
        )
    )


def triangle_intersect(v0, v1, v2, ray_origin, ray_direction) -> bool:
    """R: spatial.py:186-233 (Möller–Trumbore)."""
    epsilon = 0.0000001
    h = np.cross(ray_direction, v2 - v0)
    a = np.dot(v1 - v0, h)
    if -epsilon < a < epsilon:
        return False
    f = 1.0 / a
    s = ray_origin - v0
    u = f * np.dot(s, h)
    if u < 0.0 or u > 1.0:
        return False
    q = np.cross(s, v1 - v0)
    v = f * np.dot(ray_direction, q)
    if v < 0.0 or u + v > 1.0:
        return False
    t = f * np.dot(v2 - v0, q)
    return bool(t > epsilon)


def _resolution(points) -> float:
    """R: spatial.py:94-98."""
    from scipy.spatial import cKDTree

    distances, _ = cKDTree(points).query(points, k=2)
    return np.min(distances[:, 1])


def _distance_km_to_resolution(distance_km, lam_points, global_points) -> float:
    """R: spatial.py:101-108."""
    if isinstance(distance_km, (int, float)):
        return distance_km / R_EARTH_KM
    return _resolution({"lam": lam_points, "global": global_points, None: global_points}[distance_km])


def cutout_mask(lats, lons, global_lats, global_lons, cropping_distance=2.0, neighbours=5, min_distance_km=None,
                max_distance_km=None):
    """R: spatial.py:294-440."""
    # R: spatial.py:334-337 (a str distance such as "lam" fails the comparison itself: TypeError)
    assert cropping_distance >= 0.0, "cropping_distance must be non-negative"
    assert min_distance_km is None or min_distance_km >= 0.0, "min_distance_km must be non-negative"
    assert max_distance_km is None or max_distance_km >= 0.0, "max_distance_km must be non-negative"
    assert neighbours > 0, "neighbours must be positive"
    from scipy.spatial import cKDTree

    assert global_lats.ndim == 1 and global_lons.ndim == 1 and lats.ndim == 1 and lons.ndim == 1
    assert global_lats.shape == global_lons.shape and lats.shape == lons.shape
    north, south, east, west = np.amax(lats), np.amin(lats), np.amax(lons), np.amin(lons)
    effective_cropping_distance = cropping_distance
    if max_distance_km is not None:
        max_lat = max(abs(north), abs(south))
        R_earth_at_lat = R_EARTH_KM * np.cos(np.deg2rad(max_lat))
        L_1_degree_arc_length_km = R_earth_at_lat * RADIAN
        max_distance_degrees = max_distance_km / L_1_degree_arc_length_km
        effective_cropping_distance = max(cropping_distance, 1.1 * max_distance_degrees)
    mask = cropping_mask(
        global_lats, global_lons, np.min([90.0, north + effective_cropping_distance]), west - effective_cropping_distance,
        np.max([-90.0, south - effective_cropping_distance]), east + effective_cropping_distance,
    )
    global_points = np.array(latlon_to_xyz(global_lats[mask], global_lons[mask])).transpose()
    lam_points = np.array(latlon_to_xyz(lats, lons)).transpose()
    min_distance = _distance_km_to_resolution(min_distance_km, lam_points, global_points)
    distances, indices = cKDTree(lam_points).query(global_points, k=neighbours)
    zero = np.array([0.0, 0.0, 0.0])
    inside_lam = []
    for global_point, distance, index in zip(global_points, distances, indices):
        inside = False
        for j in range(neighbours):
            inside = triangle_intersect(
                lam_points[index[j]], lam_points[index[(j + 1) % neighbours]], lam_points[index[(j + 2) % neighbours]],
                zero, global_point,
            )
            if inside:
                break
        close = np.min(distance) <= min_distance
        too_far = False
        if max_distance_km is not None:
            too_far = np.min(distance) > (max_distance_km / R_EARTH_KM)
        inside_lam.append(inside or close or too_far)
    too_far_mask = False
    if isinstance(max_distance_km, (int, float)):
        too_far_mask = ~mask.copy()
    mask[mask] = inside_lam
    mask[too_far_mask] = True
    return ~mask


def thinning_mask(lats, lons, global_lats, global_lons, cropping_distance=2.0):
    """R: spatial.py:443-503."""
    from scipy.spatial import cKDTree

    north, south, east, west = np.amax(lats), np.amin(lats), np.amax(lons), np.amin(lons)
    mask = cropping_mask(
        global_lats, global_lons, np.min([90.0, north + cropping_distance]), west - cropping_distance,
        np.max([-90.0, south - cropping_distance]), east + cropping_distance,
    )
    global_points = np.array(latlon_to_xyz(global_lats[mask], global_lons[mask])).transpose()
    points = np.array(latlon_to_xyz(lats, lons)).transpose()
    _, indices = cKDTree(points).query(global_points, k=1)
    return indices


def global_on_lam_mask(lats, lons, global_lats, global_lons, distance_km=None):
    """R: spatial.py:506-536."""
    from scipy.spatial import cKDTree

    global_points = np.array(latlon_to_xyz(global_lats, global_lons)).transpose()
    lam_points = np.array(latlon_to_xyz(lats, lons)).transpose()
    distance = _distance_km_to_resolution(distance_km, lam_points, global_points)
    indices = cKDTree(global_points).query_ball_point(lam_points, distance)
    return np.array(sorted(set(i for sublist in indices for i in sublist)))


# --------------------------------------------------------------------------------
# multi-input per-point statements (MatchingFieldsFilter family)
# --------------------------------------------------------------------------------
def snow_depth_m(snow_depth, snow_density):
    """R: filters/fields/snow_depth_m.py:42."""
    return 1000.0 * snow_depth / snow_density


def snow_cover(snow_depth, snow_density):
    """R: filters/fields/snow_cover.py:34-39."""
    tmp1 = (1000 * snow_depth) / snow_density
    tmp2 = np.clip(snow_density, 100, 400)
    sc = np.clip(np.tanh((4000 * tmp1) / tmp2), 0, 1)
    sc[sc > 0.99] = 1.0
    return sc


def cos_sin(data, degrees: bool = False):
    """R: filters/fields/cos_sin_from_rad.py:78-79; cos_sin_mean_wave_direction.py:72-76 (degrees)."""
    if degrees:
        data = np.deg2rad(data)
    return np.cos(data), np.sin(data)


def direction_from_cos_sin(cos_values, sin_values, degrees: bool = False):
    """R: cos_sin_from_rad.py:100; cos_sin_mean_wave_direction.py:97-99 (degrees, wrapped to [0, 360))."""
    d = np.arctan2(sin_values, cos_values)
    if degrees:
        d = np.rad2deg(d)
        d = np.where(d >= 360, d - 360, d)
        d = np.where(d < 0, d + 360, d)
    return d


def xy_to_polar(u, v):
    """R: filters/fields/uv_to_ddff.py:93-97 calls ``earthkit.meteo.wind.array.xy_to_polar(u, v, convention="meteo")`` — third-party
    (earthkit-meteo >= 0.4.1, absent here).  Its published definition, restated: speed = hypot(u, v); direction = the direction the
    wind blows FROM, clockwise from north, ``mod(270 - atan2(v, u) * 180 / pi, 360)``.  Pinned by the reference's own vectors
    (tests/field_filters/test_uv_to_ddff.py:24-43, ``np.allclose``), not bit for bit."""
    speed = np.hypot(u, v)
    direction = np.mod(270.0 - np.arctan2(v, u) * (180.0 / np.pi), 360.0)
    return speed, direction.astype(speed.dtype, copy=False)


def polar_to_xy(speed, direction):
    """R: uv_to_ddff.py:121-125 calls ``earthkit.meteo.wind.array.polar_to_xy(speed, direction, convention="meteo")``: with
    a = (270 - direction) * pi / 180, u = speed * cos(a), v = speed * sin(a).  Pinned like ``xy_to_polar``."""
    a = (270.0 - direction) * (np.pi / 180.0)
    a = a.astype(np.asarray(speed).dtype, copy=False)
    return speed * np.cos(a), speed * np.sin(a)


def w_to_wz(w, t, q, level):
    """R: filters/fields/w_to_wz.py:97-99."""
    rho = (100 * level) / (287 * t * (1 + 0.61 * q) + 1e-8)
    return (-1.0 / (rho * G + 1e-8)) * w


def wz_to_w(wz, t, q, level):
    """R: w_to_wz.py:124-126."""
    rho = (100 * level) / (287 * t * (1 + 0.61 * q) + 1e-8)
    return -1.0 * rho * G * wz


def sum_fields(arrays):
    """R: filters/fields/sum.py:109-116 — in-place accumulation in order of appearance."""
    s = None
    for c in arrays:
        c = np.asarray(c).flatten()
        if s is None:
            s = c
        else:
            s += c
    return s


def interval_difference(current, previous):
    """R: filters/fields/accum_to_interval.py:98 ``fl[i].to_numpy() - fl[i - 1].to_numpy()``."""
    return current - previous


# ---- humidity conversions: earthkit-meteo's thermo.array functions, restated -----------------------------------------------------
# R: filters/fields/dewpoint.py:62-71, q_to_r.py:72-82 and q_height.py:117-142 call ``earthkit.meteo.thermo[.array]`` — third-party
# (earthkit-meteo >= 0.4.1, pyproject.toml:40; absent here).  Its published definitions, restated: the IFS saturation formulas
# es(T) = c1 exp(c3 (T - T0) / (T - c4)) over water (17.502, 32.19) and ice (22.587, -0.7) with c1 = 611.21 Pa, T0 = 273.16 K; the
# default "mixed" phase blends them with the liquid fraction ((T - Ti) / (T0 - Ti))^2, Ti = T0 - 23, held at 0 below Ti and 1 above T0;
# epsilon = Rd / Rv = 287.0597 / 461.5250.  Pinned by the reference's own vectors (tests/field_filters/test_dewpoint.py:25-29,
# test_pressure_level_humidity.py:27-40) at those tests' np.allclose — not bit for bit.
MET_C1, MET_T0, MET_TI = 611.21, 273.16, 273.16 - 23.0
MET_EPSILON = 287.0597 / 461.5250


def es_water(t):
    return MET_C1 * np.exp(17.502 * (t - MET_T0) / (t - 32.19))


def es_ice(t):
    return MET_C1 * np.exp(22.587 * (t - MET_T0) / (t - (-0.7)))


def es_mixed(t):
    """saturation_vapour_pressure(t) with its default phase "mixed"."""
    alpha = np.minimum(1.0, (np.maximum(MET_TI, np.minimum(MET_T0, t)) - MET_TI) * (1.0 / (MET_T0 - MET_TI)) ) ** 2
    return alpha * es_water(t) + (1.0 - alpha) * es_ice(t)


def dewpoint_from_relative_humidity(relative_humidity, temperature):
    """R: dewpoint.py:61-63 (the zero guard, EPS = 1e-4) -> thermo.dewpoint_from_relative_humidity: e = r es_water(t) / 100, then the
    inverse of the water formula."""
    r = np.array(relative_humidity, copy=True)
    r[r == 0] = 1.0e-4
    with np.errstate(all="ignore"):
        lnes = np.log(r * es_water(temperature) / 100.0 / MET_C1)
        return (32.19 * lnes - 17.502 * MET_T0) / (lnes - 17.502)


def relative_humidity_from_dewpoint(dewpoint, temperature):
    """R: dewpoint.py:71 -> thermo.relative_humidity_from_dewpoint: both pressures over water."""
    with np.errstate(all="ignore"):
        return 100.0 * es_water(dewpoint) / es_water(temperature)


def relative_humidity_from_specific_humidity(temperature, q, pressure):
    """R: q_to_r.py:73 / q_height.py:117 -> thermo.relative_humidity_from_specific_humidity: e = p q / (eps + eps (1/eps - 1) q)."""
    with np.errstate(all="ignore"):
        e = (pressure * q) / (MET_EPSILON + (MET_EPSILON * (1.0 / MET_EPSILON - 1.0)) * q)
        return 100.0 * e / es_mixed(temperature)


def specific_humidity_from_relative_humidity(temperature, r, pressure):
    """R: q_to_r.py:79 / q_height.py:138 -> thermo.specific_humidity_from_relative_humidity: q = eps e / (p - (1 - eps) e), NaN where
    p - e < 1e-4 (specific_humidity_from_vapour_pressure's guard)."""
    with np.errstate(all="ignore"):
        e = r * es_mixed(temperature) / 100.0
        v = np.asarray(pressure - (1.0 - MET_EPSILON) * e).copy()
        v[np.asarray(pressure - e) < 1.0e-4] = np.nan
        return MET_EPSILON * e / v


# ---- OPERA radar composites (R: filters/fields/rodeo_opera_preprocessing.py, rodeo_opera_clipping.py) -----------------------
def opera_clip_variable(variable, max_value):
    """R: rodeo_opera_preprocessing.py:34-37 — two boolean-mask assignments: a NaN fails both tests and stays, and so does -0.0."""
    variable = np.array(variable, copy=True)
    variable[variable < 0] = 0
    variable[variable >= max_value] = max_value
    return variable


def opera_clipping(tp, quality, max_total_precipitation=10000):
    """R: rodeo_opera_clipping.py:92-98 — both variables clipped (the quality index to MAX_QI = 1), then tp / FACTOR_TP (1000)."""
    return opera_clip_variable(tp, max_total_precipitation) / 1000, opera_clip_variable(quality, 1)


def opera_preprocessing(tp, quality, mask, max_total_precipitation=10000):
    """R: rodeo_opera_preprocessing.py:83-87 (mask codes 1 = no data, 2 = undetected, 3 = inf) then :190-195 (the same clipping, no division)."""
    tp, quality, mask = np.array(tp, copy=True), np.array(quality, copy=True), np.asarray(mask)
    tp[mask == 1] = np.nan
    tp[mask == 2] = 0
    tp[mask == 3] = np.nan
    quality[mask == 2] = 0
    return opera_clip_variable(tp, max_total_precipitation), opera_clip_variable(quality, 1)


# ---- ORAS6 sea-ice cleaning (R: filters/fields/oras6_clipping.py:19-21, 172-215) ---------------------------------------------
ORAS6_PUNY = 1e-5
ORAS6_MINTF = 271.15 - ORAS6_PUNY
ORAS6_TF = 273.15
ORAS6_FIELDS = ("siue", "sivn", "siconc", "icesalt", "sihc", "snhc", "sipf", "sitemptop", "sntemp", "snvol", "sivol", "sialb", "vasit", "tos")
ORAS6_OUTPUT_ORDER = ("siconc", "siue", "sivn", "icesalt", "sihc", "snhc", "sipf", "sitemptop", "sntemp", "snvol", "sivol", "sialb", "vasit", "tos")


def oras6_clipping(**arrays):
    """R: oras6_clipping.py:172-215 on the 14 named arrays; returns them in the order the reference yields them (:217-230)."""
    a = {name: np.array(arrays[name], copy=True) for name in ORAS6_FIELDS}
    with np.errstate(all="ignore"):
        if np.nanmax(a["sntemp"]) < 100:  # :190-191 — a snow temperature archived in Celsius
            a["sntemp"] = a["sntemp"] + ORAS6_TF
    mask = a["siconc"] <= ORAS6_PUNY
    for name in ("siue", "sivn", "icesalt", "sihc", "snhc", "sipf", "snvol", "sivol", "sialb"):
        a[name][mask] = 0
    for name in ("sitemptop", "sntemp", "vasit"):
        a[name][mask] = ORAS6_TF
    for name in ("sihc", "snhc"):
        a[name][a[name] >= -ORAS6_PUNY] = 0
    a["tos"][a["tos"] <= ORAS6_MINTF] = ORAS6_MINTF
    return [(name, a[name]) for name in ORAS6_OUTPUT_ORDER]


# ---- land parameters from soil / vegetation classes (R: filters/fields/land_parameters.py:20-52, 55-72) ----------------------
SOIL_TABLE = {  # class: (theta_pwp, theta_cap)
    0: (0.0, 0.0), 1: (0.059, 0.244), 2: (0.151, 0.347), 3: (0.133, 0.383), 4: (0.279, 0.448), 5: (0.335, 0.541), 6: (0.267, 0.663),
    7: (0.151, 0.347),
}
VEGETATION_TABLE = {  # class: (veg_rsmin, veg_cov, veg_z0m)
    0: (250.0, 0.0, 0.013), 1: (125.0, 0.9, 0.25), 2: (80.0, 0.85, 0.1), 3: (395.0, 0.9, 2.0), 4: (320.0, 0.9, 2.0), 5: (215.0, 0.9, 2.0),
    6: (320.0, 0.99, 2.0), 7: (100.0, 0.7, 0.5), 8: (250.0, 0.0, 0.013), 9: (45.0, 0.5, 0.03), 10: (110.0, 0.9, 0.5), 11: (45.0, 0.1, 0.03),
    12: (0.0, 0.0, 0.0013), 13: (130.0, 0.6, 0.25), 14: (0.0, 0.0, 0.0001), 15: (0.0, 0.0, 0.0001), 16: (230.0, 0.5, 0.5),
    17: (110.0, 0.4, 0.1), 18: (180.0, 0.9, 1.50), 19: (175.0, 0.9, 1.1), 20: (150.0, 0.6, 0.02),
}


def crosswalk(classes, table):
    """R: land_parameters.py:71 — one float64 array per table column, ``table[x][column]`` for every x of the class array in turn; a
    class that is not a key (a fraction, NaN, beyond the table) raises KeyError as there."""
    n_columns = len(table[0])
    return [np.array([table[x][column] for x in classes]) for column in range(n_columns)]


def filter_accum_to_interval(fields: list[dict], *, variables, zero_left: bool = True) -> list[dict]:
    """R: accum_to_interval.py:73-101 — per (param, level, levelType) group sorted by valid_datetime."""
    variables = set(variables)
    groups: dict[tuple, list[dict]] = {}
    for f in fields:
        groups.setdefault((f.get("param"), f.get("level"), f.get("levelType")), []).append(f)
    for k, fl in groups.items():
        groups[k] = sorted(fl, key=lambda x: x["valid_datetime"])
    out = []
    for (param_name, _, _), fl in groups.items():
        if param_name not in variables or len(fl) == 0:
            out.extend(fl)
            continue
        if zero_left:
            g = dict(fl[0])
            g["values"] = np.zeros_like(np.asarray(fl[0]["values"]))
            out.append(g)
        else:
            out.append(fl[0])
        for i in range(1, len(fl)):
            g = dict(fl[i])
            g["values"] = interval_difference(np.asarray(fl[i]["values"]), np.asarray(fl[i - 1]["values"]))
            out.append(g)
    return out
