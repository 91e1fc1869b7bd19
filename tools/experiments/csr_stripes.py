#!/usr/bin/env python3
"""General CSR kernel on column stacks: box-average coarsening (row length follows the latitude: ~20 entries at the poles, ~200 at the
equator), ragged rows of 3-4 and of 9-16 entries — run once per library build to compare the tile -> XCD mappings."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402
from per_level_programs import launches  # noqa: E402


def main():
    graft.load_package()
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = 137
    src, tgt = lookup("o1280"), lookup("0.25")
    n, nt = len(src["latitudes"]), len(tgt["latitudes"])
    print("library:", native.lib_path(), flush=True)
    plans = {}
    for deg in (1.0, 0.5):
        g = lookup([deg, deg])
        n_lon = int(round(360 / deg))
        cell = (np.rint((90.0 - src["latitudes"]) / deg).astype(np.int64) * n_lon + np.mod(np.rint(src["longitudes"] / deg).astype(np.int64), n_lon))
        order = np.argsort(cell, kind="stable")
        counts = np.bincount(cell, minlength=len(g["latitudes"]))
        ptr = np.concatenate([[0], np.cumsum(counts)])
        data = (1.0 / np.maximum(counts, 1))[cell[order]]
        plans[f"box average -> {deg} deg (rows of ~{int(counts.mean())})"] = (GatherPlan(n, len(g["latitudes"]), csr=(data, order.astype(np.int32), ptr)),
                                                                               lambda B, m=len(g["latitudes"]): L * B * (n + m) + n * (4 + B) + 4 * m)
    idx64, w64 = interp.knn_inverse_distance(src, lookup([1.0, 1.0]), k=64)
    m1 = len(lookup([1.0, 1.0])["latitudes"])
    plans["64 nearest neighbours -> 1 deg (as CSR)"] = (GatherPlan(n, m1, csr=(w64.reshape(-1), idx64.reshape(-1).astype(np.int32), np.arange(m1 + 1) * 64)),
                                                         lambda B: L * B * (int(np.unique(idx64).size) + m1) + idx64.size * (4 + B) + 4 * m1)
    idx, w = interp.knn_inverse_distance(src, tgt, k=4)
    keep = (np.arange(idx.size) % 9 != 0).reshape(idx.shape)
    plans["ragged(3-4) -> 0.25 deg"] = (GatherPlan(n, nt, csr=(w[keep], idx[keep], np.concatenate([[0], np.cumsum(keep.sum(axis=1))]))),
                                         lambda B: L * B * (int(np.unique(idx[keep]).size) + nt) + int(keep.sum()) * (4 + B) + 4 * nt)
    for tdt, B, tag in ((torch.float32, 4, "f32"), (torch.float64, 8, "f64")):
        x = bench.synth_stack(src, L, tdt, dev, 0, COLUMNS)
        for name, (plan, alg) in plans.items():
            ms = launches(lambda: plan.apply(x))
            print(f"{tag} csr {name:44s} {ms:7.3f} ms  {alg(B) / (ms * 1e-3) / 8e12:.3f}", flush=True)
        del x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
