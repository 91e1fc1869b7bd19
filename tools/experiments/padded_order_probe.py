#!/usr/bin/env python3
"""Ragged rows of 9-16 entries padded to k = 16 (index -1), float64: natural order against column blocks, interleaved and repeated —
`profiles/r03_kernel_bench.json` holds ONE 4.40 ms reading for the ordered float64 case next to 2.13 ms natural; is it the kernel or the box?"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def launches(fn, steps=20, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) for a, b in evs])


def main():
    graft.load_package()
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.gather import GatherPlan, target_order_for
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = 137
    src, tgt = lookup("o1280"), lookup("0.25")
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    idx16, w16 = interp.knn_inverse_distance(src, tgt, k=16, device=True, ties="index")
    keep = np.random.default_rng(16).random(idx16.shape) < 0.75
    keep[:, :9] = True
    indptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))])
    matrix = dict(matrix_data=w16[keep], matrix_indices=idx16[keep], matrix_indptr=indptr, matrix_shape=(n_tgt, n_src))
    order = target_order_for(tgt["latitudes"], tgt["longitudes"], 16)
    for tdt, B, tag in ((torch.float64, 8, "f64"), (torch.float32, 4, "f32")):
        x = bench.synth_stack(src, L, tdt, dev, 0, COLUMNS)
        alg = L * B * (int(np.unique(idx16[keep]).size) + n_tgt) + int(keep.sum()) * (4 + B) + 4 * n_tgt
        natural, ordered = GatherPlan.from_matrix(matrix), GatherPlan.from_matrix(matrix)
        ordered.order_targets(order)
        full, full_o = GatherPlan(n_src, n_tgt, index=idx16, weights=w16), GatherPlan(n_src, n_tgt, index=idx16, weights=w16)
        full_o.order_targets(order)
        out = natural.apply(x)
        assert torch.equal(out.data, ordered.apply(x).data)
        for rep in range(3):
            for name, plan, nbytes in (("padded natural", natural, alg), ("padded column blocks", ordered, alg),
                                       ("full k=16 natural", full, None), ("full k=16 column blocks", full_o, None)):
                ms = launches(lambda: plan.apply(x))
                print(f"{tag} rep {rep} {name:26s} median {np.median(ms):.3f} ms  min {ms.min():.3f}  max {ms.max():.3f}" +
                      ("" if nbytes is None else f"  {nbytes / (np.median(ms) * 1e-3) / 8e12:.3f}"), flush=True)
        idx4, w4 = idx16[:, :4], w16[:, :4] / w16[:, :4].sum(axis=1, keepdims=True)
        keep4 = (np.arange(idx4.size) % 9 != 0).reshape(idx4.shape)
        ptr4 = np.concatenate([[0], np.cumsum(keep4.sum(axis=1))])
        short = GatherPlan.from_matrix(dict(matrix_data=w4[keep4], matrix_indices=idx4[keep4], matrix_indptr=ptr4, matrix_shape=(n_tgt, n_src)))
        for rep in range(3):
            ms = launches(lambda: short.apply(x))
            print(f"{tag} rep {rep} ragged(3-4) padded to 4      median {np.median(ms):.3f} ms  min {ms.min():.3f}  max {ms.max():.3f}", flush=True)
        del x, natural, ordered, full, full_o, out, short
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
