#!/usr/bin/env python3
"""Round-4 experiment: how many longitude bands the column-block order should have on a target grid other than 0.25 degree.
O1280 -> 0.1 degree (3600 x 1801 = 6.48 M targets), k = 8, 137 float32 levels, real scattered rows (GatherPlan.order_targets).
With workgroups dealt to the 8 XCDs in contiguous ranges, 4 bands give every XCD half a band (pole to equator), 8 bands a whole one;
10 bands (what a width of 360 points gives here) do not line up with the XCD ranges.

    python tools/experiments/band_count.py
"""

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
    src, tgt = lookup("o1280"), lookup([0.1, 0.1])
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    lon = np.mod(np.asarray(tgt["longitudes"]), 360.0)
    for k in (8, 16):
        idx, w = interp.knn_inverse_distance(src, tgt, k=k, device=True, ties="index")
        x = bench.synth_stack(src, L, torch.float32, dev, 0, COLUMNS)
        alg = bench.algorithmic_bytes(L, 4, int(np.unique(idx).size), n_tgt, k)
        plans = {}
        for n_bands in (1, 2, 3, 4, 6, 8, 10, 12, 16):
            plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
            if n_bands > 1:
                band = np.minimum((lon * (n_bands / 360.0)).astype(np.int64), n_bands - 1)
                plan.order_targets(np.argsort(band, kind="stable").astype(np.int32))
            plans[n_bands] = plan
        out = plans[1].apply(x)
        ref = out.data.clone()
        results = {}
        for rnd in range(5):
            for n_bands, plan in plans.items():
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    plan.apply(x, out=out)
                b.record()
                torch.cuda.synchronize()
                if rnd == 0:
                    assert torch.equal(out.data, ref), n_bands
                else:
                    results.setdefault(n_bands, []).append(a.elapsed_time(b) / 5)
        for n_bands, v in results.items():
            med = float(np.median(v))
            print(f"k={k:2d} {n_bands:2d} band(s): median {med:7.4f} ms  min {min(v):7.4f}  frac {alg / med / 1e9 / 8:.3f}", flush=True)
        del x, out, ref, plans
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
