#!/usr/bin/env python3
"""Fixed k beyond 16 (k-NN with 32 / 64 neighbours): the tiled run-time-k ELL kernel against the general CSR kernel on the same table, natural
order and column blocks (O1280 -> 1 degree, 137 levels)."""
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
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.gather import GatherPlan, column_block_order
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = 137
    src, tgt = lookup("o1280"), lookup([0.5, 0.5])
    n, nt = len(src["latitudes"]), len(tgt["latitudes"])
    idx64, w64 = interp.knn_inverse_distance(src, tgt, k=64)
    order = column_block_order(tgt["latitudes"], tgt["longitudes"])
    for tdt, B, tag in ((torch.float32, 4, "f32"), (torch.float64, 8, "f64")):
        x = bench.synth_stack(src, L, tdt, dev, 0, COLUMNS)
        for k in (24, 32, 64):
            idx, w = idx64[:, :k], w64[:, :k] / w64[:, :k].sum(axis=1, keepdims=True)
            alg = bench.algorithmic_bytes(L, B, int(np.unique(idx).size), nt, k)
            res = []
            for kind in ("ell", "csr"):
                for ordered in (False, True):
                    plan = (GatherPlan(n, nt, index=idx, weights=w) if kind == "ell" else
                            GatherPlan(n, nt, csr=(w.reshape(-1), idx.reshape(-1).astype(np.int32), np.arange(nt + 1) * k)))
                    if ordered:
                        plan.order_targets(order)
                    ms = launches(lambda: plan.apply(x), steps=10)
                    res.append(f"{kind}{' ordered' if ordered else ''} {ms:.3f} ms {alg / (ms * 1e-3) / 8e12:.3f}")
                    del plan
            print(f"{tag} k={k}: " + " | ".join(res), flush=True)
        del x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
