#!/usr/bin/env python3
"""Round-3 experiment: long rows (k = 16) re-fetch a third of their source columns because the patches of vertically adjacent
targets overlap and a whole target row of sources (1440 targets x ~6.4 distinct columns x 560 B = 5-13 MB) does not survive in an
XCD's 4 MB L2 until the next row comes by.  Traversing the target grid in COLUMN BLOCKS of W targets (each block top to bottom)
shortens that reuse distance to W x 3.6 KB.  Emulated from outside by permuting the rows of the index / weight table (the output
rows come out in the same permuted order — the writes stay sequential, as a block-aware kernel's would be within a block).

    python tools/experiments/column_blocks.py
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


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    graft.load_package()
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS, Stack

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = 137
    src_grid, tgt_grid = lookup("o1280"), lookup("0.25")
    n_src, n_tgt = len(src_grid["latitudes"]), len(tgt_grid["latitudes"])
    n_rows, n_cols = 721, 1440
    idx16, w16 = interp.knn_inverse_distance(src_grid, tgt_grid, k=16, device=True, ties="index")
    for tdt, npdt, B, tag in ((torch.float32, np.float32, 4, "f32"), (torch.float64, np.float64, 8, "f64")):
        x = bench.synth_stack(src_grid, L, tdt, dev, 0, COLUMNS)
        out = Stack.empty(n_tgt, L, tdt, dev, COLUMNS)
        for k in (16, 8, 4):
            alg = bench.algorithmic_bytes(L, B, int(np.unique(idx16[:, :k]).size), n_tgt, k)
            wk = w16[:, :k] / w16[:, :k].sum(axis=1, keepdims=True)
            for W in (1440, 720, 480, 360, 240, 160, 96, 48):
                t = np.arange(n_tgt).reshape(n_rows, n_cols)
                perm = np.concatenate([t[:, c0:c0 + W].reshape(-1) for c0 in range(0, n_cols, W)])
                ik = torch.from_numpy(np.ascontiguousarray(idx16[perm, :k]).astype(np.int32)).to(dev)
                wd = torch.from_numpy(np.ascontiguousarray(wk[perm]).astype(npdt)).to(dev)
                ms = timeit(lambda: native.regrid_ell(x.data, out.data, ik, wd, n_src=n_src, n_tgt=n_tgt, k=k, n_lev=L, src_pitch=x.pitch,
                                                      out_pitch=out.pitch, layout=COLUMNS))
                print(f"{tag} k={k:2d} column blocks of {W:4d} targets: {ms:7.4f} ms  frac {alg / ms / 1e9 / 8:.3f}", flush=True)
        del x, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
