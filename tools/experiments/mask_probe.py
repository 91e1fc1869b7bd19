#!/usr/bin/env python3
"""apply_mask over a 137-level O1280 column stack (the by-value kernel with a point mask): time and parity, for an A/B of library builds
(ATX_LIBRARY=...): the mask byte per lane against ONE scalar fetch per wave (-DATX_PW_MASK_WAVE=1)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from anemoi_transform_amd import native  # noqa: E402
from anemoi_transform_amd.stack import COLUMNS, Stack  # noqa: E402
from relayout_probe import timeit  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n_pts, n_lev = 6599680, 137
    print(f"# library: {native.lib_path()}")
    pm = (torch.rand(n_pts, device=dev) < 0.3).to(torch.uint8)
    pm = torch.cat([pm, torch.zeros(8, dtype=torch.uint8, device=dev)])
    for dt in (torch.float32, torch.float64):
        x = Stack.empty(n_pts, n_lev, dt, dev, COLUMNS, zero=True)
        x.data[:, :n_lev].normal_(270.0, 15.0)
        y = x.new_like()
        nbytes = 2 * n_pts * n_lev * x.data.element_size() + n_pts
        for name, stages in (("apply_mask (COPY + mask)", [[(native.OP_COPY, 1, 0.0, 0.0)] * n_lev]),
                             ("convert then apply_mask", [[(native.OP_AFFINE, 0, 1.0, -273.15)] * n_lev, [(native.OP_COPY, 1, 0.0, 0.0)] * n_lev]),
                             ("affine, no mask", [[(native.OP_AFFINE, 0, 1.0, -273.15)] * n_lev])):
            prog = native.level_program(stages, dev)
            uses = any(e[1] for st in stages for e in st)
            kw = dict(n_pts=n_pts, n_lev=n_lev, x_pitch=x.pitch, y_pitch=y.pitch, layout=COLUMNS, prog=prog, n_stage=len(stages), point_mask=pm if uses else None)
            native.pointwise_stack(x.data, y.data, **kw)
            want = x.data[:, :n_lev] * 1.0 if name.startswith("apply") else x.data[:, :n_lev] * 1.0 + (-273.15)
            if uses:
                want = torch.where(pm[:n_pts, None] != 0, torch.full_like(want, float("nan")), want)
            ok = torch.equal(torch.nan_to_num(y.data[:, :n_lev], nan=-7.0), torch.nan_to_num(want, nan=-7.0))
            ms = timeit(lambda: native.pointwise_stack(x.data, y.data, **kw))
            print(f"{str(dt).split('.')[-1]:8s} {name:28s} {ms:8.4f} ms  frac {nbytes / ms / 1e6 / 8000:.3f}  {'ok' if ok else 'DIFFER'}")
        del x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
