"""Index / weight precompute for the regrid filter (CPU, one-off per grid pair).

The reference builds its gather indices with ``scipy.spatial.cKDTree`` on
unit-sphere coordinates (R: spatial.py:132-167, 587-635) and reads interpolation
matrices written by ``anemoi-transform make-regrid-file`` as ``.npz`` CSR
triplets (R: commands/make-regrid-file.py:150-160, filters/fields/regrid.py:281-290).
Both stay on the CPU here (one-off, seconds); their OUTPUTS
feed the HIP gather kernels.  The external ``mir`` binary the reference shells
out to for bilinear matrices is not available offline, so this module also
generates k-NN inverse-distance and bilinear matrices in the same npz format.
"""

from __future__ import annotations

import logging
from collections import OrderedDict
from typing import Any

import numpy as np

from .grids import gaussian_latitudes, octahedral_row_lengths

__all__ = [
    "unit_sphere_xyz",
    "nearest_grid_points",
    "nearest_grid_points_device",
    "device_knn",
    "knn_inverse_distance",
    "ell_to_csr",
    "csr_uniform_k",
    "bilinear_octahedral",
    "bilinear_rows",
    "save_matrix_npz",
    "load_matrix_npz",
]


LOG = logging.getLogger(__name__)
_knn_engine: str | None = None


def knn_engine() -> str:
    """``"ckdtree"`` (default: the reference's own builder) or ``"device"`` (``atx_knn_*`` on the GPU with
    equidistant candidates settled by cKDTree — the same table; set with ``set_knn_engine`` or ``ATX_KNN=device``)."""
    import os

    return _knn_engine or os.environ.get("ATX_KNN", "ckdtree")


def set_knn_engine(name: str | None) -> None:
    global _knn_engine
    if name not in (None, "ckdtree", "device"):
        raise ValueError(f"unknown k-NN engine {name!r}")
    _knn_engine = name


def unit_sphere_xyz(latitudes: np.ndarray, longitudes: np.ndarray) -> np.ndarray:
    """``[N, 3]`` Cartesian coordinates on the unit sphere (R: spatial.py:132-167)."""
    phi = np.deg2rad(np.asarray(latitudes, dtype=np.float64))
    lda = np.deg2rad(np.asarray(longitudes, dtype=np.float64))
    cos_phi = np.cos(phi)
    return np.array((cos_phi * np.cos(lda), cos_phi * np.sin(lda), np.sin(phi))).transpose()


MAX_DEVICE_K = 16  # atx_knn_query keeps up to 17 neighbours: these 16 plus the look-ahead one of the tie detection


# ---- remembered k-NN tables ---------------------------------------------------------------------------------------------------
# `cKDTree(src).query(tgt, k)` (R: spatial.py:628-632) is a pure function of the two point sets, k and the distance bound, and it
# is the expensive part of building a `regrid(method="nearest")` filter or a k-NN matrix: 2.4 s for O1280 -> 0.25 degree on the
# MI355X box's host, of which the tree build is 1.5 s — and a job builds the same filter for every variable / date it handles.
# Tables are therefore remembered per (source hash, target hash, k, bound, tie order): in the process (LRU, bounded in bytes)
# and on disk in the MIR-matrix npz layout (R: regrid.py:281-290: matrix_data = the chord distances, matrix_indices,
# matrix_indptr, matrix_shape) under ATX_CACHE_DIR (default ~/.cache/anemoi-transform-amd; "off" disables the files); the host
# cKDTree that settles equidistant candidates for the device search is remembered per source grid as well.
_TABLES: "OrderedDict[tuple, tuple[np.ndarray, np.ndarray]]" = OrderedDict()
_TABLES_MAX_BYTES = 1 << 30
_TREES: "OrderedDict[str, Any]" = OrderedDict()
_TREES_MAX = 2
_DISK_MIN_ENTRIES = 100_000  # smaller tables are rebuilt faster than a file is found
_cache_stats = {"memory_hits": 0, "disk_hits": 0, "misses": 0, "trees_built": 0}
KNN_TABLE_FORMAT = 2  # bump when the file layout OR the device search's tie / ordering logic changes: older files are then ignored
_cache_dir_announced = False


def _library_version() -> int:
    """``ATX_VERSION`` of the loaded libatx.so, 0 when the library has not been loaded (host-only use)."""
    try:
        from . import native

        return int(native._lib.atx_version()) if native._lib is not None else 0
    except Exception:  # noqa: BLE001 - the version is a label in the file, never a reason to fail a table build
        return 0


def points_hash(*arrays: np.ndarray) -> str:
    """128-bit content hash of float64 coordinate arrays (xxh3 when the module is there: 17 ms for O1280; blake2b otherwise)."""
    try:
        import xxhash

        h = xxhash.xxh3_128()
    except ImportError:  # pragma: no cover - depends on the installation
        import hashlib

        h = hashlib.blake2b(digest_size=16)
    for a in arrays:
        a = np.ascontiguousarray(a, dtype=np.float64)
        h.update(str(a.shape).encode())
        h.update(a.data)
    return h.hexdigest()


def knn_cache_dir() -> str | None:
    import os

    d = os.environ.get("ATX_CACHE_DIR")
    if d is not None and d.strip().lower() in ("", "0", "off", "none", "false"):
        return None
    if d is None:
        d = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "anemoi-transform-amd")
    return os.path.join(d, "knn")


def knn_cache_clear(disk: bool = False) -> None:
    _TABLES.clear()
    _TREES.clear()
    for k in _cache_stats:
        _cache_stats[k] = 0
    d = knn_cache_dir()
    if disk and d:
        import glob
        import os

        for f in glob.glob(os.path.join(d, "knn-*.npz")):
            os.remove(f)


def knn_cache_info() -> dict:
    return dict(_cache_stats, tables=len(_TABLES), bytes=sum(i.nbytes + d.nbytes for i, d in _TABLES.values()), trees=len(_TREES),
                directory=knn_cache_dir())


def host_tree(src_xyz: np.ndarray, key: str | None = None):
    """``cKDTree(src_xyz)`` remembered per source point set (the build is 60 % of a query on O1280)."""
    from scipy.spatial import cKDTree

    key = key or points_hash(src_xyz)
    tree = _TREES.get(key)
    if tree is None:
        tree = cKDTree(src_xyz)
        _cache_stats["trees_built"] += 1
        _TREES[key] = tree
        while len(_TREES) > _TREES_MAX:
            _TREES.popitem(last=False)
    else:
        _TREES.move_to_end(key)
    return tree


def _remembered_table(src_lat, src_lon, tgt_lat, tgt_lon, k: int, max_distance, order: str, compute, producer: str = "host"):
    """``(indices [n, k] int64, distances [n, k] float64)`` from the process memo, the disk cache or ``compute(src_hash)``;
    the arrays handed back are the caller's own copies.

    Entries are kept per PRODUCER (``"host"``: cKDTree itself; ``"device"``: ``atx_knn_*`` with cKDTree settling the ties) and per
    ``KNN_TABLE_FORMAT``: a table the device search built is never served to the host path (or the other way round), and a
    change to the kernel's tie handling or to the file layout is a bump of that constant, which orphans the old files instead
    of trusting them.  Files also record producer, format and library version and are re-checked on load."""
    import os

    src_h, tgt_h = points_hash(src_lat, src_lon), points_hash(tgt_lat, tgt_lon)
    bound = "none" if max_distance is None else repr(float(max_distance))
    key = (src_h, tgt_h, int(k), bound, order, producer, KNN_TABLE_FORMAT)
    hit = _TABLES.get(key)
    if hit is not None:
        _TABLES.move_to_end(key)
        _cache_stats["memory_hits"] += 1
        return hit[0].copy(), hit[1].copy()
    n_src, n_tgt = int(np.size(src_lat)), int(np.size(tgt_lat))
    directory = knn_cache_dir() if n_tgt * k >= _DISK_MIN_ENTRIES else None
    path = None
    table = None
    if directory:
        path = os.path.join(directory, f"knn-v{KNN_TABLE_FORMAT}-{producer}-{order}-{src_h}-{tgt_h}-k{k}-d{bound}.npz")
        if os.path.exists(path):
            try:
                with np.load(path) as f:
                    if (tuple(f["matrix_shape"]) == (n_tgt, n_src) and f["matrix_indices"].size == n_tgt * k
                            and int(f["format"]) == KNN_TABLE_FORMAT and str(f["producer"]) == producer and str(f["tie_order"]) == order
                            and str(f["in_grid_hash"]) == src_h and str(f["out_grid_hash"]) == tgt_h):
                        table = (f["matrix_indices"].astype(np.int64).reshape(n_tgt, k), f["matrix_data"].reshape(n_tgt, k))
                        _cache_stats["disk_hits"] += 1
            except Exception:  # a damaged file is a miss (and is rewritten below)
                table = None
    if table is None:
        _cache_stats["misses"] += 1
        indices, distances = compute(src_h)
        table = (np.ascontiguousarray(indices, dtype=np.int64).reshape(n_tgt, k), np.ascontiguousarray(distances, dtype=np.float64).reshape(n_tgt, k))
        if path:
            try:
                os.makedirs(directory, exist_ok=True)
                tmp = f"{path}.{os.getpid()}.tmp.npz"
                np.savez(tmp, matrix_data=table[1].reshape(-1), matrix_indices=table[0].astype(np.int32).reshape(-1),
                         matrix_indptr=np.arange(n_tgt + 1, dtype=np.int64) * k, matrix_shape=np.array([n_tgt, n_src]),
                         kind=np.array("k-NN table: matrix_data holds chord distances on the unit sphere, not weights"),
                         in_grid_hash=np.array(src_h), out_grid_hash=np.array(tgt_h), format=np.array(KNN_TABLE_FORMAT),
                         producer=np.array(producer), tie_order=np.array(order), library_version=np.array(_library_version()))
                os.replace(tmp, path)
                global _cache_dir_announced
                if not _cache_dir_announced:
                    _cache_dir_announced = True
                    LOG.info("k-NN tables are remembered under %s (ATX_CACHE_DIR=off disables the files)", directory)
            except OSError:  # a read-only or full cache directory must not fail the filter
                pass
    _TABLES[key] = table
    total = sum(i.nbytes + d.nbytes for i, d in _TABLES.values())
    while total > _TABLES_MAX_BYTES and len(_TABLES) > 1:
        _, (i, d) = _TABLES.popitem(last=False)
        total -= i.nbytes + d.nbytes
    return table[0].copy(), table[1].copy()


def device_knn(src_xyz: np.ndarray, tgt_xyz: np.ndarray, k: int, *, ties: str = "ckdtree", max_distance: float | None = None,
               tree_key: str | None = None):
    """``(indices int64 [n, k], squared distances float64 [n, k], n_rows_resolved)`` — the k nearest rows of
    ``src_xyz`` for every row of ``tgt_xyz``, searched on the MI355X (``atx_knn_build`` / ``atx_knn_query``).

    The kernel's distances are bit-identical to cKDTree's, so its neighbour lists can differ from cKDTree's only in the
    ORDER (or, at the k-th place, the choice) of EXACTLY equidistant candidates: the kernel orders them by source index,
    cKDTree by whichever it meets first in its traversal (R: spatial.py:628 — an artefact of ``std::nth_element`` during
    its tree build, not reproducible without running that build).  ``ties="ckdtree"`` (default) makes the result
    identical to the reference's anyway: the device returns one neighbour more than asked for, rows in which two adjacent
    distances among those k+1 are equal are found on the device, and only those rows (0.14 % for O1280 -> 0.25 degree)
    are answered again by ``cKDTree(src).query`` on the host — the reference's own statement.  The host tree is built
    only if such rows exist.  ``ties="index"`` skips that and keeps the kernel's deterministic order.
    """
    import torch

    from . import native
    from . import stack as _stack

    if ties not in ("ckdtree", "index"):
        raise ValueError(f"ties must be 'ckdtree' or 'index', got {ties!r}")
    if not 1 <= k <= MAX_DEVICE_K:
        raise ValueError(f"the device k-NN search returns 1..{MAX_DEVICE_K} neighbours, got k={k}")
    dev = _stack.device()
    src_xyz = np.ascontiguousarray(src_xyz, dtype=np.float64)
    tgt_xyz = np.ascontiguousarray(tgt_xyz, dtype=np.float64)
    n_tgt = len(tgt_xyz)
    if n_tgt == 0:
        return np.zeros((0, k), dtype=np.int64), np.zeros((0, k)), 0
    ahead = 1 if ties == "ckdtree" else 0
    idx_d, d2_d = native.KnnIndex(torch.from_numpy(src_xyz).to(dev)).query(torch.from_numpy(tgt_xyz).to(dev), k + ahead)
    rows = None
    if ahead:
        # a finite distance equal to its successor: the candidates' order (or which of them is the k-th) is cKDTree's to decide
        tied = ((d2_d[:, :-1] == d2_d[:, 1:]) & torch.isfinite(d2_d[:, :-1])).any(dim=1)
        rows = torch.nonzero(tied).reshape(-1).cpu().numpy()
    indices = idx_d[:, :k].cpu().numpy().astype(np.int64)
    d2 = d2_d[:, :k].cpu().numpy()
    if rows is not None and rows.size:
        kwargs = {} if max_distance is None else {"distance_upper_bound": max_distance}
        host_d, host_i = host_tree(src_xyz, tree_key).query(tgt_xyz[rows], k=k, **kwargs)
        host_d, host_i = host_d.reshape(len(rows), k), host_i.reshape(len(rows), k)
        # cKDTree reports sqrt(d2); the squared distances are the same multiset per row, re-ordered like the indices
        found = host_i < len(src_xyz)
        diff = src_xyz[np.where(found, host_i, 0)] - tgt_xyz[rows][:, None, :]
        host_d2 = np.zeros(host_i.shape)
        for c in range(3):  # s = 0; s += dx*dx; ... — scipy's order, as in the kernel
            host_d2 = host_d2 + diff[..., c] * diff[..., c]
        indices[rows] = host_i
        d2[rows] = np.where(found, host_d2, np.inf)
    return indices, d2, 0 if rows is None else int(rows.size)


def nearest_grid_points_device(
    source_latitudes,
    source_longitudes,
    target_latitudes,
    target_longitudes,
    max_distance: float | None = None,
    num_neighbours_to_return: int = 1,
    return_distances: bool = False,
    ties: str = "ckdtree",
):
    """``nearest_grid_points`` computed on the MI355X (``atx_knn_build`` / ``atx_knn_query``).

    Same arguments and return values as R: spatial.py:587-635, and — with the default ``ties="ckdtree"`` — the same
    bits: distances are bit-identical to cKDTree's (coordinates are computed on the host exactly as the reference does,
    the kernel repeats scipy's float64 arithmetic), and rows with exactly equidistant candidates are settled by cKDTree
    itself (``device_knn``).  ``ties="index"`` orders such candidates by source index and never touches the host tree.
    """
    k = int(num_neighbours_to_return)

    def compute(src_hash):
        src = unit_sphere_xyz(source_latitudes, source_longitudes)
        tgt = unit_sphere_xyz(target_latitudes, target_longitudes)
        indices, d2, _ = device_knn(src, tgt, k, ties=ties, max_distance=max_distance, tree_key=src_hash)
        distances = np.sqrt(d2)
        if max_distance is not None:  # cKDTree: neighbours at d >= distance_upper_bound are "missing"
            missing = ~(distances < max_distance)
            indices[missing] = len(src)
            distances[missing] = np.inf
        return indices, distances

    indices, distances = _remembered_table(source_latitudes, source_longitudes, target_latitudes, target_longitudes, k, max_distance,
                                           ties, compute, producer="device")
    if k == 1:
        indices, distances = indices[:, 0], distances[:, 0]
    if return_distances:
        return indices, distances
    return indices


def nearest_grid_points(
    source_latitudes,
    source_longitudes,
    target_latitudes,
    target_longitudes,
    max_distance: float | None = None,
    num_neighbours_to_return: int = 1,
    return_distances: bool = False,
):
    """k nearest source points of every target point, by chord distance.

    Same call surface and return order as the reference (R: spatial.py:587-635):
    ``indices`` (int64, ``[Nt]`` or ``[Nt, k]``) or ``(indices, distances)``.
    With ``max_distance`` cKDTree reports ``len(source)`` for "none found"
    (R: spatial.py:630-632); the regrid filter rejects such indices before any
    gather (``native.check_indices``).
    """
    if knn_engine() == "device":
        return nearest_grid_points_device(
            source_latitudes, source_longitudes, target_latitudes, target_longitudes, max_distance=max_distance,
            num_neighbours_to_return=num_neighbours_to_return, return_distances=return_distances,
        )
    k = int(num_neighbours_to_return)

    def compute(src_hash):
        tree = host_tree(unit_sphere_xyz(source_latitudes, source_longitudes), src_hash)
        kwargs = {} if max_distance is None else {"distance_upper_bound": max_distance}
        distances, indices = tree.query(unit_sphere_xyz(target_latitudes, target_longitudes), k=k, **kwargs)
        return indices, distances

    indices, distances = _remembered_table(source_latitudes, source_longitudes, target_latitudes, target_longitudes, k, max_distance,
                                           "ckdtree", compute)
    if k == 1:  # cKDTree's own shapes: [Nt] for k = 1, [Nt, k] otherwise
        indices, distances = indices[:, 0], distances[:, 0]
    if return_distances:
        return indices, distances
    return indices


def knn_inverse_distance(in_grid: dict, out_grid: dict, k: int = 4, floor: float = 1e-12, device: bool = False,
                         ties: str = "ckdtree"):
    """k-NN inverse-distance weights ``w_j = (1/max(d_j, floor)) / sum``.

    Returns ``(idx [Nt, k] int64, w [Nt, k] float64)``.  ``device=True`` runs the neighbour
    search on the GPU (``nearest_grid_points_device``; ``ties`` as there) instead of cKDTree.
    """
    kwargs = dict(ties=ties) if device else {}
    search = nearest_grid_points_device if device else nearest_grid_points
    idx, dist = search(
        in_grid["latitudes"], in_grid["longitudes"], out_grid["latitudes"], out_grid["longitudes"],
        num_neighbours_to_return=k, return_distances=True, **kwargs,
    )
    idx = idx.reshape(len(idx), -1)
    dist = dist.reshape(len(dist), -1)
    inv = 1.0 / np.maximum(dist, floor)
    return idx, inv / inv.sum(axis=1, keepdims=True)


def ell_to_csr(idx: np.ndarray, w: np.ndarray, n_src: int) -> dict[str, np.ndarray]:
    """Fixed-k index/weight table as the CSR triplet of the MIR npz format."""
    n_tgt, k = idx.shape
    return dict(
        matrix_data=np.ascontiguousarray(w, dtype=np.float64).reshape(-1),
        matrix_indices=np.ascontiguousarray(idx, dtype=np.int32).reshape(-1),
        matrix_indptr=(np.arange(n_tgt + 1, dtype=np.int64) * k).astype(np.int32),
        matrix_shape=np.array([n_tgt, n_src]),
    )


def csr_uniform_k(indptr: np.ndarray) -> int | None:
    """Row length if every CSR row holds the same number of entries, else ``None``."""
    lengths = np.diff(indptr)
    if lengths.size and (lengths == lengths[0]).all() and lengths[0] > 0:
        return int(lengths[0])
    return None


def bilinear_rows(row_latitudes: np.ndarray, row_lengths: np.ndarray, out_grid: dict) -> dict[str, np.ndarray]:
    """Bilinear weights from a ROW-STRUCTURED global grid (rows of constant latitude, north to south, row ``r`` holding
    ``row_lengths[r]`` equally spaced longitudes starting at 0 — octahedral / classic reduced / full Gaussian and regular
    lat-lon grids) to arbitrary target points.

    Two bracketing rows x two bracketing longitudes per row (periodic), linear in longitude on each row, then linear in
    latitude; above the first / below the last row the nearest row is used alone.  Always 4 entries per target (zero
    weights are kept) so the matrix is fixed-k.
    """
    lats = np.asarray(row_latitudes, dtype=np.float64)  # north -> south
    nlon = np.asarray(row_lengths, dtype=np.int64)
    if lats.ndim != 1 or lats.shape != nlon.shape or len(lats) < 1 or np.any(np.diff(lats) >= 0) or np.any(nlon < 1):
        raise ValueError("rows must be given north to south with strictly decreasing latitudes and positive lengths")
    row_start = np.concatenate([[0], np.cumsum(nlon)[:-1]])
    tlat = np.asarray(out_grid["latitudes"], dtype=np.float64)
    tlon = np.mod(np.asarray(out_grid["longitudes"], dtype=np.float64), 360.0)

    # first row whose latitude is <= target latitude (rows descend)
    south = np.searchsorted(-lats, -tlat, side="left")
    north = south - 1
    south = np.clip(south, 0, len(lats) - 1)
    north = np.clip(north, 0, len(lats) - 1)
    span = lats[north] - lats[south]
    w_north = np.where(span > 0, (tlat - lats[south]) / np.where(span > 0, span, 1.0), 1.0)
    w_north = np.clip(w_north, 0.0, 1.0)

    idx = np.empty((len(tlat), 4), dtype=np.int64)
    w = np.empty((len(tlat), 4), dtype=np.float64)
    for slot, (row, w_row) in enumerate(((north, w_north), (south, 1.0 - w_north))):
        m = nlon[row]
        x = tlon * (m / 360.0)
        j0 = np.floor(x).astype(np.int64)
        frac = x - j0
        j0 = np.mod(j0, m)
        j1 = np.mod(j0 + 1, m)
        idx[:, 2 * slot] = row_start[row] + j0
        idx[:, 2 * slot + 1] = row_start[row] + j1
        w[:, 2 * slot] = w_row * (1.0 - frac)
        w[:, 2 * slot + 1] = w_row * frac
    return ell_to_csr(idx, w, int(nlon.sum()))


def bilinear_octahedral(n: int, out_grid: dict) -> dict[str, np.ndarray]:
    """``bilinear_rows`` from the octahedral grid ``O<n>`` (BASELINE config 2: O96 -> 1 degree)."""
    return bilinear_rows(gaussian_latitudes(2 * n), octahedral_row_lengths(n), out_grid)


def save_matrix_npz(path: str, matrix: dict[str, np.ndarray], in_grid: dict, out_grid: dict) -> None:
    """Write the regrid-matrix npz the reference reads (R: make-regrid-file.py:150-160)."""
    np.savez(
        path,
        matrix_data=matrix["matrix_data"],
        matrix_indices=matrix["matrix_indices"],
        matrix_indptr=matrix["matrix_indptr"],
        matrix_shape=matrix["matrix_shape"],
        in_latitudes=in_grid["latitudes"],
        in_longitudes=in_grid["longitudes"],
        out_latitudes=out_grid["latitudes"],
        out_longitudes=out_grid["longitudes"],
    )


def load_matrix_npz(path: str) -> dict[str, Any]:
    """R: regrid.py:281."""
    return dict(np.load(path))
