#!/usr/bin/env python3
"""Round-4 experiment: does a 2-D visiting order of the targets do better for long rows (k = 16) than the column blocks of round 3?

Emulated like tools/experiments/column_blocks.py — the rows of the index / weight table permuted, the output written in the permuted
(sequential) order — for: natural order, column blocks of 360, tiles of TH x TW targets (row-major inside a tile, tiles row-major
inside bands of TH rows), and the Morton (Z) order of the whole 721 x 1440 grid.

    python tools/experiments/tile_orders.py
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
from column_blocks import timeit  # noqa: E402


def tiles(n_rows, n_cols, th, tw):
    t = np.arange(n_rows * n_cols).reshape(n_rows, n_cols)
    parts = []
    for r0 in range(0, n_rows, th):
        for c0 in range(0, n_cols, tw):
            parts.append(t[r0:r0 + th, c0:c0 + tw].reshape(-1))
    return np.concatenate(parts)


def morton(n_rows, n_cols):
    r, c = np.meshgrid(np.arange(n_rows), np.arange(n_cols), indexing="ij")

    def spread(v):
        v = v.astype(np.uint64)
        out = np.zeros_like(v)
        for b in range(12):
            out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(2 * b)
        return out

    code = (spread(r.reshape(-1)) << np.uint64(1)) | spread(c.reshape(-1))
    return np.argsort(code, kind="stable")


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
    orders = {"natural": np.arange(n_tgt), "column blocks of 360": tiles(n_rows, n_cols, n_rows, 360)}
    for th, tw in ((2, 2), (2, 4), (4, 4), (4, 8), (8, 8), (16, 16), (32, 32), (64, 64), (8, 360), (32, 360)):
        orders[f"tiles {th} x {tw}"] = tiles(n_rows, n_cols, th, tw)
    orders["Morton"] = morton(n_rows, n_cols)
    for name, perm in orders.items():
        assert np.array_equal(np.sort(perm), np.arange(n_tgt)), name
    for tdt, npdt, B, tag in ((torch.float32, np.float32, 4, "f32"), (torch.float64, np.float64, 8, "f64")):
        x = bench.synth_stack(src_grid, L, tdt, dev, 0, COLUMNS)
        out = Stack.empty(n_tgt, L, tdt, dev, COLUMNS)
        for k in (16, 8):
            alg = bench.algorithmic_bytes(L, B, int(np.unique(idx16[:, :k]).size), n_tgt, k)
            wk = w16[:, :k] / w16[:, :k].sum(axis=1, keepdims=True)
            for name, perm in orders.items():
                ik = torch.from_numpy(np.ascontiguousarray(idx16[perm, :k]).astype(np.int32)).to(dev)
                wd = torch.from_numpy(np.ascontiguousarray(wk[perm]).astype(npdt)).to(dev)
                ms = timeit(lambda: native.regrid_ell(x.data, out.data, ik, wd, n_src=n_src, n_tgt=n_tgt, k=k, n_lev=L, src_pitch=x.pitch,
                                                      out_pitch=out.pitch, layout=COLUMNS))
                print(f"{tag} k={k:2d} {name:22s}: {ms:7.4f} ms  frac {alg / ms / 1e9 / 8:.3f}", flush=True)
        del x, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
