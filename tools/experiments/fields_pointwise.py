#!/usr/bin/env python3
"""Per-point programs on FIELD-MAJOR stacks (the reference's own array order, what a ctypes binder passes with ATX_FIELDS):
137 fields of O1280, fraction of 8 TB/s on 2 x stack bytes."""
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
    from anemoi_transform_amd import native
    from anemoi_transform_amd.stack import FIELDS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n, L = 6599680, 137
    print("library:", native.lib_path(), flush=True)
    for tdt, B, tag in ((torch.float32, 4, "f32"), (torch.float64, 8, "f64")):
        x = torch.rand((L, n), dtype=tdt, device=dev) * 30 + 270
        y = torch.empty_like(x)
        mask = (torch.arange(n, device=dev) % 3 == 0).to(torch.uint8)
        kw = dict(n_pts=n, n_lev=L, x_pitch=n, y_pitch=n, layout=FIELDS)
        A, CP, CL = native.OP_AFFINE, native.OP_COPY, native.OP_CLIP
        cases = {
            "affine": ([[(A, 0, 2.0, 1.0)] * L], False),
            "a scale per level": ([[(A, 0, 1.0 + 0.001 * l, -273.15) for l in range(L)]], False),
            "two stages": ([[(native.OP_MUL, 0, 9.80665, 0.0)] * L, [(A, 0, 1.0, -273.15)] * L], False),
            "clip": ([[(CL, 0, 275.0, 295.0)] * L], False),
            "apply_mask": ([[(CP, 1, 0.0, 0.0)] * L], True),
            "every third level converted": ([[(A, 0, 1.0, -273.15) if l % 3 == 0 else (CP, 0, 0.0, 0.0) for l in range(L)]], False),
        }
        for name, (stages, uses_mask) in cases.items():
            prog = native.level_program(stages, dev)
            extra = {"point_mask": mask} if uses_mask else {}
            for place, dst in (("out-of-place", y), ("in-place", x)):
                ms = launches(lambda: native.pointwise_stack(x, dst, prog=prog, n_stage=len(stages), **extra, **kw))
                print(f"{tag} fields {name:30s} {place:12s} {ms:7.3f} ms  {2 * n * L * B / (ms * 1e-3) / 8e12:.3f}", flush=True)
        # the other entry points on field-major stacks
        ms = launches(lambda: native.reduce_stack(x, native.RED_MINMAX, n_pts=n, n_lev=L, pitch=n, layout=FIELDS))
        print(f"{tag} fields reduce min+max {ms:7.3f} ms  {n * L * B / (ms * 1e-3) / 8e12:.3f}", flush=True)
        z = torch.empty_like(x)
        ang = torch.rand((L, n), dtype=tdt, device=dev) * 6.28  # wave directions in radians (R: cos_sin_from_rad.py:73-76 checks the range)
        for name, op, ins, outs in (("difference", native.COMB_SUB, [x, y], [z]), ("cos_sin", native.COMB_COS_SIN, [ang], [y, z])):
            ms = launches(lambda: native.combine_stack(op, ins, outs, n_pts=n, n_lev=L, pitch=n, layout=FIELDS))
            print(f"{tag} fields combine {name:12s} {ms:7.3f} ms  {(len(ins) + len(outs)) * n * L * B / (ms * 1e-3) / 8e12:.3f}", flush=True)
        half = torch.empty((68, n), dtype=tdt, device=dev)
        ms = launches(lambda: native.select_levels(x, half, list(range(0, 136, 2)), n_pts=n, n_src_lev=L, src_pitch=n, dst_pitch=n, layout=FIELDS))
        print(f"{tag} fields select 68 of 137 levels {ms:7.3f} ms  {2 * 68 * n * B / (ms * 1e-3) / 8e12:.3f}", flush=True)
        del x, y, z, half, ang
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
