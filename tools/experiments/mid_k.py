#!/usr/bin/env python3
"""Fixed k between 4 and 8 against the algorithmic roofline: natural order and the library's policy (column blocks from k = 5 on).
Round 2: the runtime-k tiled kernel; round 3: compile-time k on the direct kernel up to 8."""
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
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = 137
    src, tgt = lookup("o1280"), lookup("0.25")
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    idx8, w8 = interp.knn_inverse_distance(src, tgt, k=8, device=True, ties="index")
    for tdt, B, tag in ((torch.float32, 4, "f32"), (torch.float64, 8, "f64")):
        x = bench.synth_stack(src, L, tdt, dev, 0, COLUMNS)
        for k in (4, 5, 6, 8):
            idx, w = idx8[:, :k], w8[:, :k] / w8[:, :k].sum(axis=1, keepdims=True)
            plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
            alg = bench.algorithmic_bytes(L, B, int(np.unique(idx).size), n_tgt, k)
            ms, _ = bench.time_launches(lambda: plan.apply(x), 10, 2)
            from anemoi_transform_amd.gather import target_order_for

            order = target_order_for(tgt["latitudes"], tgt["longitudes"], k)
            ms_o = None
            if order is not None:
                plan.order_targets(order)
                ms_o, _ = bench.time_launches(lambda: plan.apply(x), 10, 2)
            print(f"{tag} fixed k={k}: {ms:.3f} ms  {alg / (ms * 1e-3) / 8e12:.3f} of 8 TB/s on algorithmic bytes ({alg / 1e9:.2f} GB)" +
                  ("" if ms_o is None else f"   | targets in column blocks (the policy): {ms_o:.3f} ms  {alg / (ms_o * 1e-3) / 8e12:.3f}"), flush=True)
        del x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
