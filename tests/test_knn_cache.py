"""The remembered k-NN tables of interp.py: a second `regrid(method="nearest")` over the same grid pair pays neither the cKDTree
build nor the query (R: spatial.py:587-635 is a pure function of its arguments), in the process and across processes through
files in the MIR-matrix npz layout."""

from __future__ import annotations

import glob
import os

import numpy as np
import pytest

from anemoi_transform_amd import interp
from anemoi_transform_amd.grids import lookup


@pytest.fixture()
def cache(tmp_path, monkeypatch):
    monkeypatch.setenv("ATX_CACHE_DIR", str(tmp_path))
    monkeypatch.setattr(interp, "_DISK_MIN_ENTRIES", 1)
    interp.knn_cache_clear()
    yield tmp_path
    interp.knn_cache_clear()


def _grids():
    return lookup("o16"), lookup([10.0, 10.0])


def test_second_call_builds_no_tree_and_equals_ckdtree(cache):
    from scipy.spatial import cKDTree

    src, tgt = _grids()
    args = (src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"])
    i1, d1 = interp.nearest_grid_points(*args, num_neighbours_to_return=4, return_distances=True)
    assert interp.knn_cache_info()["trees_built"] == 1 and interp.knn_cache_info()["misses"] == 1
    i2, d2 = interp.nearest_grid_points(*args, num_neighbours_to_return=4, return_distances=True)
    info = interp.knn_cache_info()
    assert info["trees_built"] == 1 and info["memory_hits"] == 1 and info["misses"] == 1
    want_d, want_i = cKDTree(interp.unit_sphere_xyz(*args[:2])).query(interp.unit_sphere_xyz(*args[2:]), k=4)
    for i, d in ((i1, d1), (i2, d2)):
        assert i.dtype == np.int64 and np.array_equal(i, want_i) and np.array_equal(d, want_d)
    # the caller owns what it gets: scribbling on it does not reach the next caller
    i2[:] = -7
    i3 = interp.nearest_grid_points(*args, num_neighbours_to_return=4)
    assert np.array_equal(i3, want_i)
    # k = 1 keeps cKDTree's 1-D shape and is its own entry; so is a distance bound (missing neighbours = len(source), inf)
    one = interp.nearest_grid_points(*args)
    _, want_one = cKDTree(interp.unit_sphere_xyz(*args[:2])).query(interp.unit_sphere_xyz(*args[2:]), k=1)  # (ties: not column 0 of k = 4)
    assert one.shape == (len(tgt["latitudes"]),) and np.array_equal(one, want_one)
    bi, bd = interp.nearest_grid_points(*args, max_distance=0.05, num_neighbours_to_return=4, return_distances=True)
    wd, wi = cKDTree(interp.unit_sphere_xyz(*args[:2])).query(interp.unit_sphere_xyz(*args[2:]), k=4, distance_upper_bound=0.05)
    assert np.array_equal(bi, wi) and np.array_equal(bd, wd) and (bi == len(src["latitudes"])).any()
    assert interp.knn_cache_info()["trees_built"] == 1  # one source grid, one tree, however many tables


def test_tables_persist_in_the_matrix_npz_layout(cache):
    src, tgt = _grids()
    args = (src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"])
    i1, d1 = interp.nearest_grid_points(*args, num_neighbours_to_return=3, return_distances=True)
    files = glob.glob(os.path.join(str(cache), "knn", "knn-*.npz"))
    assert len(files) == 1 and "-k3-" in files[0]
    m = interp.load_matrix_npz(files[0])  # R: regrid.py:281-290 keys
    n_tgt, n_src = len(tgt["latitudes"]), len(src["latitudes"])
    assert tuple(m["matrix_shape"]) == (n_tgt, n_src) and m["matrix_indices"].dtype == np.int32
    assert np.array_equal(m["matrix_indptr"], np.arange(n_tgt + 1) * 3) and np.array_equal(m["matrix_data"].reshape(n_tgt, 3), d1)
    # a new process (memory forgotten, files kept): no tree, no query
    interp.knn_cache_clear()
    i2, d2 = interp.nearest_grid_points(*args, num_neighbours_to_return=3, return_distances=True)
    info = interp.knn_cache_info()
    assert info["disk_hits"] == 1 and info["trees_built"] == 0 and info["misses"] == 0
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    # a damaged file is a miss, and is replaced
    interp.knn_cache_clear()
    with open(files[0], "wb") as f:
        f.write(b"not an npz")
    i3 = interp.nearest_grid_points(*args, num_neighbours_to_return=3)
    assert np.array_equal(i3, i1) and interp.knn_cache_info()["misses"] == 1
    assert tuple(interp.load_matrix_npz(files[0])["matrix_shape"]) == (n_tgt, n_src)


def test_files_can_be_switched_off(cache, monkeypatch):
    monkeypatch.setenv("ATX_CACHE_DIR", "off")
    src, tgt = _grids()
    interp.nearest_grid_points(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"])
    assert interp.knn_cache_dir() is None and not glob.glob(os.path.join(str(cache), "**", "*.npz"), recursive=True)
    assert interp.knn_cache_info()["tables"] == 1  # the process memo still works


def test_regrid_filter_construction_reuses_the_table(cache, monkeypatch):
    """Two `regrid(method="nearest")` filters over one grid pair: the second one's plan comes from the memo (no tree, no query)."""
    import native_double
    from anemoi_transform_amd.fields import fieldlist_from_dicts
    from anemoi_transform_amd.filters import create_filter_by_name

    native_double.install(monkeypatch)
    src, tgt = _grids()
    rng = np.random.default_rng(3)
    specs = [{"param": "t", "levelist": l, "values": rng.standard_normal(len(src["latitudes"])), "latitudes": src["latitudes"],
              "longitudes": src["longitudes"], "valid_datetime": "2020-01-01T00:00:00Z"} for l in (1, 2)]
    outs = []
    for _ in range(2):
        regrid = create_filter_by_name("regrid", in_grid="o16", out_grid=[10.0, 10.0], method="nearest")
        outs.append([f.to_numpy(flatten=True) for f in regrid.forward(fieldlist_from_dicts(specs))])
    info = interp.knn_cache_info()
    assert info["trees_built"] == 1 and info["misses"] == 1 and info["memory_hits"] == 1
    assert all(np.array_equal(a, b) for a, b in zip(*outs))


def test_tables_are_kept_per_producer_and_format(cache, monkeypatch):
    """A table the device search built is never served to the host path (or back), and a file of another format is ignored
    (the advisor's round-3 finding: a retuned kernel must not leave trusted stale files behind)."""
    src, tgt = _grids()
    args = (src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"])
    calls = []

    def compute_marked(marker):
        def compute(src_hash):
            calls.append(marker)
            n = len(tgt["latitudes"])
            return np.full((n, 2), marker, dtype=np.int64), np.full((n, 2), float(marker))
        return compute

    host = interp._remembered_table(*args, 2, None, "ckdtree", compute_marked(1), producer="host")
    device = interp._remembered_table(*args, 2, None, "ckdtree", compute_marked(2), producer="device")
    assert calls == [1, 2] and host[0][0, 0] == 1 and device[0][0, 0] == 2  # same grids, k, bound, tie order: two entries
    assert interp._remembered_table(*args, 2, None, "ckdtree", compute_marked(3), producer="host")[0][0, 0] == 1
    files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(str(cache), "knn", "knn-*.npz")))
    assert len(files) == 2 and f"knn-v{interp.KNN_TABLE_FORMAT}-device-ckdtree-" in files[0] and f"knn-v{interp.KNN_TABLE_FORMAT}-host-ckdtree-" in files[1]
    with np.load(os.path.join(str(cache), "knn", files[0])) as f:
        assert str(f["producer"]) == "device" and int(f["format"]) == interp.KNN_TABLE_FORMAT and "library_version" in f
    # a new format number: memory and files of the old one are not consulted
    interp.knn_cache_clear()
    monkeypatch.setattr(interp, "KNN_TABLE_FORMAT", interp.KNN_TABLE_FORMAT + 1)
    assert interp._remembered_table(*args, 2, None, "ckdtree", compute_marked(4), producer="host")[0][0, 0] == 4
    # a file whose recorded producer does not match its name (copied by hand) is a miss
    interp.knn_cache_clear()
    newest = [f for f in glob.glob(os.path.join(str(cache), "knn", "knn-*.npz")) if f"knn-v{interp.KNN_TABLE_FORMAT}-host" in f]
    assert len(newest) == 1
    os.replace(newest[0], newest[0].replace("-host-", "-device-"))
    assert interp._remembered_table(*args, 2, None, "ckdtree", compute_marked(5), producer="device")[0][0, 0] == 5
