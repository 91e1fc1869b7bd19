#!/usr/bin/env python3
"""Per-point programs whose operators DIFFER from level to level (a scale per level, a packed surface stack with one variable
converted, two such stages, with a point mask) on 137 levels of O1280: fraction of 8 TB/s on 2 x stack bytes.
Run once per library (ATX_LIBRARY) to compare kernels."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def launches(fn, steps=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


def main():
    graft.load_package()
    from anemoi_transform_amd import native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS, Stack

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    src = lookup("o1280")
    n = len(src["latitudes"])
    print("library:", native.lib_path(), flush=True)
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.gather import GatherPlan

    tgt = lookup("0.25")
    idx, w = interp.knn_inverse_distance(src, tgt, k=4)
    plan = GatherPlan(n, len(tgt["latitudes"]), index=idx, weights=w)
    keep = (np.arange(idx.size) % 9 != 0).reshape(idx.shape)
    csr = GatherPlan(n, len(tgt["latitudes"]), csr=(w[keep], idx[keep], np.concatenate([[0], np.cumsum(keep.sum(axis=1))])))
    for L in (137, 13):
        for tdt, B, tag in ((torch.float32, 4, "f32"), (torch.float64, 8, "f64")):
            x = bench.synth_stack(src, L, tdt, dev, 0, COLUMNS)
            y = x.new_like()
            mask = (torch.arange(n, device=dev) % 3 == 0).to(torch.uint8)
            kw = dict(n_pts=n, n_lev=L, x_pitch=x.pitch, y_pitch=y.pitch, layout=COLUMNS)
            A, M, CP, CL = native.OP_AFFINE, native.OP_MUL, native.OP_COPY, native.OP_CLIP
            cases = {
                "a scale per level": [[(A, 0, 1.0 + 0.001 * l, -273.15) for l in range(L)]],
                "two stages, a scale per level each": [[(M, 0, 1.0 + 0.001 * l, 0.0) for l in range(L)], [(A, 0, 1.0, -0.5 * l) for l in range(L)]],
                "groups of 4 levels share a scale": [[(A, 0, 1.0 + 0.001 * (l // 4), -273.15) for l in range(L)]],
                "every third level converted, the rest copied": [[(A, 0, 1.0, -273.15) if l % 3 == 0 else (CP, 0, 0.0, 0.0) for l in range(L)]],
                "scale / clip alternating": [[(A, 0, 1.0 + 0.001 * l, 1.0) if l % 2 else (CL, 0, 0.0, 300.0) for l in range(L)]],
                "a scale per level + point mask on all": [[(A, 1, 1.0 + 0.001 * l, -273.15) for l in range(L)]],
            }
            for name, stages in cases.items():
                prog = native.level_program(stages, dev)
                uses_mask = any(o[1] for st in stages for o in st)
                extra = {"point_mask": mask} if uses_mask else {}
                for place, dst in (("out-of-place", y), ("in-place", x)):
                    ms = launches(lambda: native.pointwise_stack(x.data, dst.data, prog=prog, n_stage=len(stages), **extra, **kw))
                    frac = 2 * n * L * B / (ms * 1e-3) / 8e12
                    print(f"L={L:3d} {tag} {name:46s} {place:12s} {ms:7.3f} ms  {frac:.3f}", flush=True)
            if L == 137:  # the same programs as the epilogue of the regrid (O1280 -> 0.25 degrees, k = 4)
                for name in ("a scale per level", "two stages, a scale per level each", "groups of 4 levels share a scale",
                             "every third level converted, the rest copied"):
                    stages = cases[name]
                    prog = native.level_program(stages, dev)
                    ms0 = launches(lambda: plan.apply(x))
                    ms = launches(lambda: plan.apply(x, prog=prog, n_stage=len(stages)))
                    print(f"L={L:3d} {tag} regrid k=4 + {name:46s} {ms:7.3f} ms  (no epilogue {ms0:.3f} ms)", flush=True)
                # the same epilogues on the general CSR kernel (ragged rows 3-4, forced off the padded route) and a general operator (tiled kernel)
                prog = native.level_program(cases["a scale per level"], dev)
                ms0 = launches(lambda: csr.apply(x))
                ms = launches(lambda: csr.apply(x, prog=prog, n_stage=1))
                print(f"L={L:3d} {tag} regrid_csr ragged(3-4) + a scale per level                       {ms:7.3f} ms  (no epilogue {ms0:.3f} ms)", flush=True)
                prog = native.level_program(cases["scale / clip alternating"], dev)
                ms = launches(lambda: plan.apply(x, prog=prog, n_stage=1))
                print(f"L={L:3d} {tag} regrid k=4 + scale / clip alternating (general operators: tiled kernel) {ms:7.3f} ms", flush=True)
            del x, y
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
