#!/usr/bin/env python3
"""lnsp_to_sp (exp) and sp_to_lnsp (log) as single-stage per-point programs over 137 levels of O1280: time and fraction of 8 TB/s."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from per_level_programs import launches  # noqa: E402


def main():
    graft.load_package()
    from anemoi_transform_amd import native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    src = lookup("o1280")
    n, L = len(src["latitudes"]), 137
    print("library:", native.lib_path(), flush=True)
    for tdt, B, tag in ((torch.float32, 4, "f32"), (torch.float64, 8, "f64")):
        x = bench.synth_stack(src, L, tdt, dev, 0, COLUMNS)  # ~ 200 .. 320: log is fine, exp of it overflows f32 -> scale first
        x.data.mul_(0.03)
        y = x.new_like()
        kw = dict(n_pts=n, n_lev=L, x_pitch=x.pitch, y_pitch=y.pitch, layout=COLUMNS)
        for name, stages in (("affine", [[(native.OP_AFFINE, 0, 2.0, 1.0)] * L]), ("exp", [[(native.OP_EXP, 0, 0.0, 0.0)] * L]),
                             ("log", [[(native.OP_LOG, 0, 0.0, 0.0)] * L]),
                             ("log then exp", [[(native.OP_LOG, 0, 0.0, 0.0)] * L, [(native.OP_EXP, 0, 0.0, 0.0)] * L])):
            prog = native.level_program(stages, dev)
            ms = launches(lambda: native.pointwise_stack(x.data, y.data, prog=prog, n_stage=len(stages), **kw))
            print(f"{tag} {name:14s} {ms:7.3f} ms  {2 * n * L * B / (ms * 1e-3) / 8e12:.3f}", flush=True)
        del x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
