#!/usr/bin/env python3
"""Headline launch (O1280 -> 0.25 degree, k = 4, 137 levels f32) against the COLUMN PITCH of source and output: does aligning
columns to 64 / 128 bytes (pitch 144 / 160 elements instead of 140) buy anything?"""
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
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = 137
    src, tgt = lookup("o1280"), lookup("0.25")
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    idx64, w64 = interp.knn_inverse_distance(src, tgt, k=4, device=True, ties="index")
    for tdt, npdt, B, tag in ((torch.float32, np.float32, 4, "f32"), (torch.float64, np.float64, 8, "f64")):
        idx = torch.from_numpy(idx64.astype(np.int32)).to(dev)
        w = torch.from_numpy(w64.astype(npdt)).to(dev)
        alg = bench.algorithmic_bytes(L, B, int(np.unique(idx64).size), n_tgt, 4)
        base = (L * B + 15) // 16 * 16 // B
        results = {}
        for rounds in range(3):  # interleaved
            for sp in (base, (L * B + 63) // 64 * 64 // B, (L * B + 127) // 128 * 128 // B):
                for op in (base, (L * B + 63) // 64 * 64 // B):
                    x = torch.randn(n_src, sp, dtype=tdt, device=dev)
                    out = torch.empty(n_tgt, op, dtype=tdt, device=dev)
                    ms, _ = bench.time_launches(lambda: native.regrid_ell(x, out, idx, w, n_src=n_src, n_tgt=n_tgt, k=4, n_lev=L, src_pitch=sp, out_pitch=op,
                                                                          layout=COLUMNS), 20, 3)
                    results.setdefault((sp, op), []).append(ms)
                    del x, out
        for (sp, op), t in results.items():
            ms = float(np.median(t))
            print(f"{tag} source pitch {sp:3d} ({sp * B} B) output pitch {op:3d}: {ms * 1e3:7.1f} us  {alg / (ms * 1e-3) / 8e12:.3f}", flush=True)


if __name__ == "__main__":
    main()
