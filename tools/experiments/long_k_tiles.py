#!/usr/bin/env python3
"""Fixed k = 8 / 16 (runtime-k tiled kernel): launch time against the tile size (targets per workgroup)."""
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
    from anemoi_transform_amd import interp, native
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
    for k in (8, 16):
        idx, w = idx16[:, :k], w16[:, :k] / w16[:, :k].sum(axis=1, keepdims=True)
        plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
        alg = bench.algorithmic_bytes(L, 4, int(np.unique(idx).size), n_tgt, k)
        line = f"k={k:2d} f32:"
        for tile in (0, 2, 4, 8, 16, 32, 64):
            native.set_tuning(tile)
            ms, _ = bench.time_launches(lambda: plan.apply(x), 10, 2)
            line += f"  tile {tile or 'auto'}: {ms:.3f} ms ({alg / (ms * 1e-3) / 8e12:.3f})"
        native.set_tuning(0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
