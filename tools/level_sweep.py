#!/usr/bin/env python3
"""Regrid k=4 / k=1 launch time against the number of levels in the stack and the tile size (targets per workgroup):
checks that the tile heuristic of atx_regrid.hip (tuned on 137 levels) holds for surface fields (1 level), pressure-level
stacks (13, 37) and 60 / 137 model levels.  Column stacks, O1280 -> 0.25 degree."""

from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    graft.load_package()
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS, Stack

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    src, tgt = lookup("o1280"), lookup("0.25")
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    idx64, w64 = interp.knn_inverse_distance(src, tgt, k=4, device=True, ties="index")
    U = {4: int(np.unique(idx64).size), 1: int(np.unique(idx64[:, 0]).size)}
    for tdt, npdt, B, tag in ((torch.float32, np.float32, 4, "f32"), (torch.float64, np.float64, 8, "f64")):
        for L in (1, 4, 13, 37, 60, 137):
            x = bench.synth_stack(src, L, tdt, dev, 0, COLUMNS)
            out = Stack.empty(n_tgt, L, tdt, dev, COLUMNS)
            for k in (4, 1):
                idx = torch.from_numpy(np.ascontiguousarray(idx64[:, :k]).astype(np.int32)).to(dev)
                w = torch.from_numpy(w64.astype(npdt)).to(dev) if k > 1 else None
                alg = bench.algorithmic_bytes(L, B, U[k], n_tgt, k)
                line = f"{tag} L={L:3d} k={k}: "
                best = None
                for tile in (0, 8, 16, 32, 64, 128, 256):
                    native.set_tuning(tile)
                    ms, _ = bench.time_launches(lambda: native.regrid_ell(x.data, out.data, idx, w, n_src=n_src, n_tgt=n_tgt, k=k, n_lev=L,
                                                                          src_pitch=x.pitch, out_pitch=out.pitch, layout=COLUMNS), 20, 3)
                    line += f" tile {tile if tile else 'auto':>4}: {ms * 1e3:7.1f} us"
                    if tile and (best is None or ms < best[1]):
                        best = (tile, ms)
                    if tile == 0:
                        auto = ms
                native.set_tuning(0)
                print(line + f"   | auto/best = {auto / best[1]:.2f} (best tile {best[0]}), auto frac of 8 TB/s on algorithmic bytes {alg / (auto * 1e-3) / 8e12:.3f}",
                      flush=True)
            del x, out
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
