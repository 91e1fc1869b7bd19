#!/usr/bin/env python3
"""Column stacks whose pitch exceeds the 16-byte-rounded level count (a C caller aligning columns to 128 bytes, a view into a wider
stack): per-point program, reduction, gather — against the tight-pitch numbers."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as graft  # noqa: E402
from per_level_programs import launches  # noqa: E402


def main():
    graft.load_package()
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    src, tgt = lookup("o1280"), lookup("0.25")
    n, nt, L = len(src["latitudes"]), len(tgt["latitudes"]), 137
    idx, w = interp.knn_inverse_distance(src, tgt, k=4)
    idx_d = torch.from_numpy(idx.astype(np.int32).reshape(-1)).to(dev)
    print("library:", native.lib_path(), flush=True)
    for tdt, B, tag in ((torch.float32, 4, "f32"), (torch.float64, 8, "f64")):
        vec = 16 // B
        tight = (L + vec - 1) // vec * vec
        w_d = torch.from_numpy(w.astype(np.float32 if B == 4 else np.float64).reshape(-1)).to(dev)
        for pitch, what in ((tight, "tight"), ((L * B + 127) // 128 * 128 // B, "128-byte aligned columns"), (tight + vec, "one spare vector")):
            x = torch.rand((n, pitch), dtype=tdt, device=dev) * 30 + 270
            y = torch.empty_like(x)
            out = torch.empty((nt, pitch), dtype=tdt, device=dev)
            kw = dict(n_pts=n, n_lev=L, x_pitch=pitch, y_pitch=pitch, layout=COLUMNS)
            uni = native.level_program([[(native.OP_AFFINE, 0, 2.0, 1.0)] * L], dev)
            per = native.level_program([[(native.OP_AFFINE, 0, 1.0 + 0.001 * l, -273.15) for l in range(L)]], dev)
            alg = 2 * n * L * B
            ms_u = launches(lambda: native.pointwise_stack(x, y, prog=uni, n_stage=1, **kw))
            ms_p = launches(lambda: native.pointwise_stack(x, y, prog=per, n_stage=1, **kw))
            ms_r = launches(lambda: native.reduce_stack(x, native.RED_MINMAX, n_pts=n, n_lev=L, pitch=pitch, layout=COLUMNS))
            ms_g = launches(lambda: native.regrid_ell(x, out, idx_d, w_d, n_src=n, n_tgt=nt, k=4, n_lev=L, src_pitch=pitch, out_pitch=pitch, layout=COLUMNS))
            print(f"{tag} pitch {pitch:4d} ({what:24s}): affine {ms_u:6.3f} ms {alg / (ms_u * 1e-3) / 8e12:.3f} | a scale per level {ms_p:6.3f} ms "
                  f"{alg / (ms_p * 1e-3) / 8e12:.3f} | min+max {ms_r:6.3f} ms {alg / 2 / (ms_r * 1e-3) / 8e12:.3f} | regrid k=4 {ms_g:6.3f} ms", flush=True)
            del x, y, out
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
