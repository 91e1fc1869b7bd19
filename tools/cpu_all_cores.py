#!/usr/bin/env python3
"""The CPU's best case for the headline workload (baseline B): the reference statement
`csr_array(k=4) @ x` (float64, R: regrid.py:310) over independent fields on W worker PROCESSES (scipy holds the GIL in
csr_matvec, threads do not scale).  Touches no GPU; bench.py runs it as a child process and copies the JSON line.

    python tools/cpu_all_cores.py --seconds 8                      # os.cpu_count() workers
    python tools/cpu_all_cores.py --sweep 16,32,64,128,256 --seconds 6   # one JSON line per worker count + the best
"""

from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_matrix = None
_fields = None


def _work(args):
    """Interpolate field `i % len(_fields)` until the deadline; returns the number of fields done."""
    i, deadline = args
    done = 0
    x = _fields[i % len(_fields)]
    while time.perf_counter() < deadline:
        _matrix @ x
        done += 1
    return done


def main():
    global _matrix, _fields
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=0,
                    help="worker processes; 0 = every core this process may use: os.cpu_count(), capped by the cgroup CPU quota "
                         "(the MI355X test boxes grant 16 of 256 host threads; more workers than that only time-slice — "
                         "profiles/r02_cpu_workers_sweep.jsonl: 16 -> 3.7e9, 32 -> 2.5e9, 256 -> 0.8e9 points/s)")
    ap.add_argument("--sweep", default="", help="comma-separated worker counts: time each, print every line, then the best again")
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--src-grid", default="o1280")
    ap.add_argument("--tgt-grid", default="0.25")
    ap.add_argument("--k", type=int, default=4)
    args = ap.parse_args()
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[var] = "1"

    import __graft_entry__ as graft

    graft.load_package()
    from scipy.sparse import csr_array

    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.interp import knn_inverse_distance

    src, tgt = lookup(args.src_grid), lookup(args.tgt_grid)
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    idx, w = knn_inverse_distance(src, tgt, k=args.k)  # host cKDTree
    indptr = (np.arange(n_tgt + 1, dtype=np.int64) * args.k).astype(np.int32)
    _matrix = csr_array((w.reshape(-1), idx.astype(np.int32).reshape(-1), indptr), shape=(n_tgt, n_src))
    rng = np.random.default_rng(20260630)
    lat, lon = np.deg2rad(src["latitudes"]), np.deg2rad(src["longitudes"])
    _fields = [280 + 30 * np.sin(lat) * np.cos(2 * lon + 0.1 * l) + rng.standard_normal(n_src) for l in range(4)]
    _matrix @ _fields[0]

    ctx = mp.get_context("fork")  # workers inherit the matrix and the fields; nothing here has touched a GPU

    def cpu_quota():
        try:  # cgroup v2 CPU quota of the box, in cores
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            return None if quota == "max" else float(quota) / float(period)
        except Exception:
            return None

    if args.workers <= 0:
        args.workers = len(os.sched_getaffinity(0))
        if cpu_quota():
            args.workers = max(1, min(args.workers, int(cpu_quota())))

    def run(workers: int) -> dict:
        with ctx.Pool(workers) as pool:
            pool.map(_work, [(i, time.perf_counter() + 0.2) for i in range(workers)])  # all workers forked and warm
            t0 = time.perf_counter()
            deadline = t0 + args.seconds
            done = sum(pool.map(_work, [(i, deadline) for i in range(workers)], chunksize=1))
            elapsed = time.perf_counter() - t0
        return {"value": done * n_tgt / elapsed, "unit": "grid-points/s", "cores": workers, "kind": "port", "fields": done,
                "seconds": elapsed, "ms_per_field_per_worker": elapsed * workers / max(done, 1) * 1e3,
                "host_logical_cores": os.cpu_count(), "usable_cores": len(os.sched_getaffinity(0)), "cpu_quota_cores": cpu_quota()}

    if args.sweep:
        results = [run(int(w)) for w in args.sweep.split(",")]
        for r in results:
            print(json.dumps(r), flush=True)
        best = max(results, key=lambda r: r["value"])
        print(json.dumps(dict(best, sweep={str(r["cores"]): r["value"] for r in results})))
    else:
        print(json.dumps(run(args.workers)))


if __name__ == "__main__":
    main()
