#!/usr/bin/env python3
"""Would a 2-D blocked traversal of the targets help long rows?  The index table's rows are permuted on the host into a Z-order
of (latitude, longitude) cells and the unchanged kernels run over it (outputs land in permuted order: timing only)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def spread(v):
    v = v.astype(np.uint64) & np.uint64(0xFFFF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF)
    v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F)
    v = (v | (v << np.uint64(2))) & np.uint64(0x33333333)
    v = (v | (v << np.uint64(1))) & np.uint64(0x55555555)
    return v


def z_order(lat, lon, cell_deg):
    iy = np.floor((90.0 - lat) / cell_deg).astype(np.int64)
    ix = np.floor(np.mod(lon, 360.0) / cell_deg).astype(np.int64)
    key = spread(iy) << np.uint64(1) | spread(ix)
    return np.argsort(key, kind="stable")


def main():
    graft.load_package()
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = 137
    src, tgt = lookup("o1280"), lookup("0.25")
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    idx16, w16 = interp.knn_inverse_distance(src, tgt, k=16, device=True, ties="index")
    x = bench.synth_stack(src, L, torch.float32, dev, 0, COLUMNS)
    for k in (4, 8, 16):
        idx, w = idx16[:, :k], w16[:, :k] / w16[:, :k].sum(axis=1, keepdims=True)
        alg = bench.algorithmic_bytes(L, 4, int(np.unique(idx).size), n_tgt, k)
        line = f"k={k:2d}:"
        for name, order in [("row-major", None)] + [(f"Z-order cells of {c} deg", z_order(tgt["latitudes"], tgt["longitudes"], c)) for c in (0.5, 1.0, 2.0, 4.0)]:
            ii, ww = (idx, w) if order is None else (idx[order], w[order])
            plan = GatherPlan(n_src, n_tgt, index=ii, weights=ww)
            ms, _ = bench.time_launches(lambda: plan.apply(x), 10, 2)
            line += f"  {name}: {ms:.3f} ms ({alg / (ms * 1e-3) / 8e12:.3f})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
