#!/usr/bin/env python3
"""Long CSR rows (box-average coarsening O1280 -> 1 degree, ~100 entries per row; k = 16 nearest neighbours): launch time of
the general CSR / runtime-k kernels against the tile size (targets per workgroup)."""

from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    graft.load_package()
    from anemoi_transform_amd import native
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = 137
    src = lookup("o1280")
    n_src = len(src["latitudes"])
    one = lookup([1.0, 1.0])
    n_one = len(one["latitudes"])
    cell = np.rint(90.0 - src["latitudes"]).astype(np.int64) * 360 + np.mod(np.rint(src["longitudes"]).astype(np.int64), 360)
    order = np.argsort(cell, kind="stable")
    counts = np.bincount(cell, minlength=n_one)
    indptr = np.concatenate([[0], np.cumsum(counts)])
    data = (1.0 / np.maximum(counts, 1))[cell[order]]
    box = GatherPlan(n_src, n_one, csr=(data, order.astype(np.int32), indptr))
    for tdt, B, tag in ((torch.float32, 4, "f32"), (torch.float64, 8, "f64")):
        x = bench.synth_stack(src, L, tdt, dev, 0, COLUMNS)
        alg = L * B * (n_src + n_one) + n_src * (4 + B) + 4 * n_one
        line = f"box average {tag}:"
        for tile in (0, 2, 4, 8, 16, 32, 64):
            native.set_tuning(tile)
            try:
                ms, _ = bench.time_launches(lambda: box.apply(x), 10, 2)
                line += f"  tile {tile or 'auto'}: {ms:.3f} ms ({alg / (ms * 1e-3) / 8e12:.3f})"
            except Exception as e:
                line += f"  tile {tile}: {type(e).__name__}"
        native.set_tuning(0)
        print(line, flush=True)
        del x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
