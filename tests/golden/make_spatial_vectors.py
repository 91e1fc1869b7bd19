#!/usr/bin/env python3
"""Golden vectors for the index / mask builders, produced by RUNNING THE REFERENCE ITSELF.

``anemoi/transform/spatial.py`` is the one slice of the reference that can execute in the build container
(SURVEY.md §8c): it needs numpy / scipy plus four names from ``earthkit.meteo.constants.constants``
(R: constants.py:11-14).  This script imports it from ``/root/reference/src`` and records what its functions return:

    nearest_grid_points   R: spatial.py:587-635   indices + chord distances, k = 1 and 4, with and without max_distance
    cutout_mask           R: spatial.py:294-440   7 option sets x 5 limited-area patches (+ the TypeError of a str distance)
    thinning_mask         R: spatial.py:443-503
    global_on_lam_mask    R: spatial.py:506-536   distance_km = number / "lam" / "global" / None
    cropping_mask         R: spatial.py:236-275   boxes that wrap the dateline / the Greenwich meridian
    regrid                R: regrid.py:380, :310  `field[..., idx]` and `csr_array(inverse-distance weights) @ field` evaluated by numpy / scipy
                                                  on the index / distance tables nearest_grid_points returned above (seeded fields)

How to run (build container only; neither the reference nor the shim travels to the GPU box — only the .npz does):

    a throw-away package providing ``earthkit.meteo.constants.constants`` with the four constants SURVEY.md §8c lists
    (R, R_earth, g, radian) must be on PYTHONPATH, in a directory OUTSIDE this repository (e.g. under /tmp);
    PYTHONPATH=<that directory> python3 tests/golden/make_spatial_vectors.py            # writes spatial_vectors.npz
    PYTHONPATH=<that directory> python3 tests/golden/make_spatial_vectors.py --check    # re-runs the reference and compares with the committed file

Inputs are either formula grids of this repository (``grids.lookup`` — stored by NAME, with a content hash, the tests
regenerate them) or small seeded arrays stored in the file.  Outputs are stored exactly (int32 indices, float64 distances,
bit-packed boolean masks).  ``tests/test_spatial_vectors.py`` holds the oracle and the host builders to this file,
``tests/test_gpu_spatial_vectors.py`` the device builders.
"""

from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "spatial_vectors.npz")
REFERENCE_SRC = os.environ.get("ATX_REFERENCE_SRC", "/root/reference/src")

# ---- the cases (shared with the tests through the manifest stored in the file) ---------------------------------------------

KNN_PAIRS = {
    # name: (source grid, target grid, distance_upper_bound for the bounded variants — chosen so that SOME neighbours are missing)
    "o32_to_5deg": ("o32", [5.0, 5.0], 0.02),
    "o96_to_1deg": ("o96", [1.0, 1.0], 0.006),
    # regular -> regular: every second target sits exactly between source points (exact ties), the pole rows are 36 copies of one point
    "reg10_to_5deg": ([10.0, 10.0], [5.0, 5.0], 0.09),
}

CUTOUT_OPTIONS = [
    {},
    {"min_distance_km": 100.0},
    {"max_distance_km": 500.0},
    {"min_distance_km": 50, "max_distance_km": 800, "cropping_distance": 1.0},
    {"neighbours": 7},
    {"neighbours": 3, "min_distance_km": 0},
    {"cropping_distance": 5.0, "neighbours": 9, "min_distance_km": 30.0},
]


def lam_patches() -> dict[str, tuple[np.ndarray, np.ndarray, str]]:
    """name -> (lats, lons, global grid name).  Seeded; regular and jittered; both longitude conventions."""
    rng = np.random.default_rng(20260704)

    def mesh(lat0, lat1, dlat, lon0, lon1, dlon, jitter=0.0):
        lats, lons = np.meshgrid(np.arange(lat0, lat1 + 1e-9, dlat), np.arange(lon0, lon1 + 1e-9, dlon), indexing="ij")
        lats, lons = lats.reshape(-1).copy(), lons.reshape(-1).copy()
        if jitter:
            lats += rng.normal(0.0, jitter, lats.shape)
            lons += rng.normal(0.0, jitter, lons.shape)
        return lats, lons

    return {
        # negative longitudes against a 0..360 global grid, across the Greenwich meridian
        "europe": (*mesh(35.0, 55.0, 0.5, -5.0, 20.0, 0.5), "o96"),
        # across the dateline, longitudes beyond 180
        "dateline": (*mesh(-20.0, 10.0, 0.75, 170.0, 195.0, 0.75, jitter=0.05), "o96"),
        # high latitudes, irregular
        "arctic": (*mesh(70.0, 88.0, 0.5, 0.0, 60.0, 1.5, jitter=0.08), "o96"),
        # southern hemisphere, irregular
        "south": (*mesh(-60.0, -40.0, 1.0, 280.0, 320.0, 1.0, jitter=0.1), "o96"),
        # an UNJITTERED patch inside a regular global grid: coincident points, exactly equidistant neighbours, degenerate triangles
        "europe_on_regular_2deg": (*mesh(36.0, 54.0, 0.5, 2.0, 18.0, 0.5), [2.0, 2.0]),
    }


CROPPING_BOXES = [  # (north, west, south, east)
    (50.0, 350.0, 30.0, 370.0),
    (10.0, -20.0, -10.0, 20.0),
    (90.0, 170.0, 60.0, 190.0),
    (-30.0, -190.0, -60.0, -170.0),
    (20.0, 100.0, -20.0, 140.0),
]


def grid_hash(grid: dict) -> str:
    h = hashlib.sha256()
    for key in ("latitudes", "longitudes"):
        h.update(np.ascontiguousarray(grid[key], dtype=np.float64).tobytes())
    return h.hexdigest()


def option_tag(options: dict) -> str:
    return "default" if not options else ",".join(f"{k}={options[k]!r}" for k in sorted(options))


def main() -> int:
    try:
        import earthkit.meteo.constants.constants as shim  # noqa: F401
    except ImportError:
        print(__doc__, file=sys.stderr)
        print("earthkit.meteo.constants.constants is not importable: put the throw-away constants package on PYTHONPATH "
              "(outside this repository)", file=sys.stderr)
        return 2
    shim_path = os.path.abspath(shim.__file__)
    assert not shim_path.startswith(ROOT + os.sep), f"the constants shim must live outside the repository, found {shim_path}"
    sys.path.insert(0, REFERENCE_SRC)
    from anemoi.transform import spatial as ref  # the reference's own module

    assert os.path.abspath(ref.__file__).startswith(os.path.abspath(REFERENCE_SRC)), ref.__file__

    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft

    graft.load_package()
    from anemoi_transform_amd.grids import lookup  # formula grids: inputs only

    import scipy

    arrays: dict[str, np.ndarray] = {}
    manifest: dict = {
        "generator": "tests/golden/make_spatial_vectors.py",
        "reference_module": "anemoi/transform/spatial.py (imported from the reference tree, run in the build container)",
        "numpy": np.__version__,
        "scipy": scipy.__version__,
        "constants": {"R_earth_km": float(ref.R_earth_km), "radian": float(ref.radian)},
        "grids": {},
        "nearest_grid_points": [],
        "cutout_mask": [],
        "cutout_mask_errors": [],
        "thinning_mask": [],
        "global_on_lam_mask": [],
        "cropping_mask": [],
    }

    def grid(spec):
        g = lookup(spec)
        name = spec if isinstance(spec, str) else "x".join(str(v) for v in spec)
        manifest["grids"][name] = {"spec": spec, "n": int(len(g["latitudes"])), "sha256": grid_hash(g)}
        return name, g

    # ---- nearest_grid_points -------------------------------------------------------------------------------------------------
    for pair, (src_spec, tgt_spec, bound) in KNN_PAIRS.items():
        src_name, src = grid(src_spec)
        tgt_name, tgt = grid(tgt_spec)
        for k in (1, 4):
            for max_distance in (None, bound):
                idx, dist = ref.nearest_grid_points(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"],
                                                    max_distance=max_distance, num_neighbours_to_return=k, return_distances=True)
                only_idx = ref.nearest_grid_points(src["latitudes"], src["longitudes"], tgt["latitudes"], tgt["longitudes"],
                                                   max_distance=max_distance, num_neighbours_to_return=k)
                assert np.array_equal(idx, only_idx)
                key = f"ngp/{pair}/k{k}/{'unbounded' if max_distance is None else 'bounded'}"
                arrays[key + "/idx"] = idx.astype(np.int32)
                arrays[key + "/dist"] = dist.astype(np.float64)
                d = dist.reshape(len(dist), -1)
                manifest["nearest_grid_points"].append({
                    "key": key, "source": src_name, "target": tgt_name, "k": k, "max_distance": max_distance,
                    "shape": list(idx.shape), "missing": int(np.sum(idx == len(src["latitudes"]))),
                    "rows_with_equal_adjacent_distances": int(np.sum((d[:, :-1] == d[:, 1:]).any(axis=1))) if k > 1 else 0,
                })

    # ---- the two regrid statements on the reference's own index tables -------------------------------------------------------------
    # R: filters/fields/regrid.py:380 `data[..., nearest_grid_points]` and :310 `csr_array(...) @ data` cannot be reached through the
    # reference's filter classes here (they import earthkit / anemoi.utils), but their INPUT — the index table — is the reference's own
    # (nearest_grid_points above), and the statements themselves are numpy / scipy calls: recorded on a seeded field so that the GPU
    # path is held to vectors whose indices and distances come from the reference (k = 4: inverse-distance weights w = (1/max(d, 1e-12)) / sum,
    # SURVEY.md §8d).
    from scipy.sparse import csr_array

    manifest["regrid"] = []
    for pair in ("o32_to_5deg", "o96_to_1deg"):
        src_spec, tgt_spec, _ = KNN_PAIRS[pair]
        src, tgt = lookup(src_spec), lookup(tgt_spec)
        n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
        seed = 20260630
        rng = np.random.default_rng(seed)
        lat, lon = np.deg2rad(src["latitudes"]), np.deg2rad(src["longitudes"])
        n_levels = 3 if n_tgt < 10000 else 1  # (one level of the larger pair keeps the file small)
        fields = np.stack([280.0 + 30.0 * np.sin(lat) * np.cos(2.0 * lon + 0.1 * l) + rng.standard_normal(n_src) for l in range(n_levels)])
        idx1 = arrays[f"ngp/{pair}/k1/unbounded/idx"].astype(np.int64)
        idx4 = arrays[f"ngp/{pair}/k4/unbounded/idx"].astype(np.int64)
        dist4 = arrays[f"ngp/{pair}/k4/unbounded/dist"]
        inv = 1.0 / np.maximum(dist4, 1e-12)
        weights = inv / inv.sum(axis=1, keepdims=True)
        matrix = csr_array((weights.reshape(-1), idx4.reshape(-1).astype(np.int32), (np.arange(n_tgt + 1) * 4).astype(np.int32)), shape=(n_tgt, n_src))
        key = f"regrid/{pair}"
        arrays[key + "/nearest"] = np.stack([f[..., idx1] for f in fields])
        arrays[key + "/knn4"] = np.stack([matrix @ f for f in fields])
        manifest["regrid"].append({"key": key, "pair": pair, "source": src_spec, "target": tgt_spec, "seed": seed, "levels": n_levels,
                                   "field": "280 + 30 sin(lat) cos(2 lon + 0.1 l) + default_rng(seed).standard_normal(n_src), level by level"})

    # ---- the limited-area builders ---------------------------------------------------------------------------------------------
    patches = lam_patches()
    for name, (lats, lons, global_spec) in patches.items():
        arrays[f"lam/{name}/lats"], arrays[f"lam/{name}/lons"] = lats, lons
        global_name, glob = grid(global_spec)
        glats, glons = glob["latitudes"], glob["longitudes"]
        for options in CUTOUT_OPTIONS:
            mask = ref.cutout_mask(lats, lons, glats.copy(), glons.copy(), **options)
            assert mask.dtype == bool and mask.shape == glats.shape
            key = f"cutout/{name}/{option_tag(options)}"
            arrays[key] = np.packbits(mask)
            manifest["cutout_mask"].append({"key": key, "lam": name, "global": global_name, "options": options, "n": int(mask.size),
                                            "kept": int(mask.sum())})
        for cropping_distance in (2.0, 0.5):
            indices = ref.thinning_mask(lats, lons, glats.copy(), glons.copy(), cropping_distance=cropping_distance)
            key = f"thinning/{name}/{cropping_distance}"
            arrays[key] = np.asarray(indices).astype(np.int32)
            manifest["thinning_mask"].append({"key": key, "lam": name, "global": global_name, "cropping_distance": cropping_distance,
                                              "n": int(len(indices))})
        for distance_km in (60.0, 25, "lam", "global", None):
            indices = ref.global_on_lam_mask(lats, lons, glats.copy(), glons.copy(), distance_km=distance_km)
            key = f"global_on_lam/{name}/{distance_km!r}"
            arrays[key] = np.asarray(indices).astype(np.int32)
            manifest["global_on_lam_mask"].append({"key": key, "lam": name, "global": global_name, "distance_km": distance_km,
                                                   "n": int(len(indices))})

    # a str distance passes _distance_km_to_resolution but not cutout_mask's own assertion (R: spatial.py:336): it raises
    lats, lons, global_spec = patches["europe"]
    glob = lookup(global_spec)
    for bad in ({"min_distance_km": "lam"}, {"min_distance_km": "global"}, {"cropping_distance": -1.0}, {"neighbours": 0}):
        try:
            ref.cutout_mask(lats, lons, glob["latitudes"], glob["longitudes"], **bad)
            raised = None
        except Exception as e:  # noqa: BLE001 - the type IS the vector
            raised = type(e).__name__
        manifest["cutout_mask_errors"].append({"lam": "europe", "global": "o96", "options": bad, "raises": raised})

    # ---- cropping_mask -----------------------------------------------------------------------------------------------------------
    _, o32 = grid("o32")
    for convention in ("0..360", "-180..180"):
        lons = o32["longitudes"] if convention == "0..360" else np.where(o32["longitudes"] >= 180.0, o32["longitudes"] - 360.0, o32["longitudes"])
        for box in CROPPING_BOXES:
            mask = ref.cropping_mask(o32["latitudes"], lons, *box)
            key = f"cropping/o32/{convention}/{box!r}"
            arrays[key] = np.packbits(mask)
            manifest["cropping_mask"].append({"key": key, "grid": "o32", "longitudes": convention, "box": list(box), "n": int(mask.size),
                                              "inside": int(mask.sum())})

    arrays["manifest"] = np.array(json.dumps(manifest, indent=1))
    if "--check" in sys.argv:  # nothing is written: what the reference returns NOW against the committed vectors
        with np.load(OUT) as committed:
            missing = sorted(set(arrays) - set(committed.files)), sorted(set(committed.files) - set(arrays))
            differ = [k for k in arrays if k in committed.files and k != "manifest" and not np.array_equal(arrays[k], committed[k])]
            old = json.loads(str(committed["manifest"]))
        for key in ("numpy", "scipy"):
            if old.get(key) != manifest[key]:
                print(f"note: the committed vectors were produced with {key} {old.get(key)}, this run uses {manifest[key]}")
        if any(missing) or differ:
            print(f"MISMATCH: only here {missing[0]}, only in the file {missing[1]}, different {differ}")
            return 1
        print(f"{OUT}: all {len(arrays) - 1} arrays equal what the reference returns in this container")
        return 0
    np.savez_compressed(OUT, **arrays)
    print(f"wrote {OUT}: {len(arrays)} arrays, {os.path.getsize(OUT) / 1e6:.2f} MB")
    for section in ("nearest_grid_points", "regrid", "cutout_mask", "thinning_mask", "global_on_lam_mask", "cropping_mask", "cutout_mask_errors"):
        print(f"  {section}: {len(manifest[section])} cases")
    return 0


if __name__ == "__main__":
    sys.exit(main())
