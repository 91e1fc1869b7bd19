#!/usr/bin/env python3
"""Per-point programs made of RUNS of levels (several variables sharing a column): the by-value runs kernel (round 4) against the routes such
programs took before — build a variant with `bash tools/build_variant.sh noruns -DATX_PW_RUNS=0 -DATX_EPI_RUNS=0` and run with
ATX_LIBRARY=anemoi-transform_amd/lib/variants/libatx_noruns.so for the other side."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from anemoi_transform_amd import native  # noqa: E402
from anemoi_transform_amd.stack import COLUMNS, Stack  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from relayout_probe import timeit  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n_pts = 6599680
    print(f"# library: {native.lib_path()}")
    aff, mul, cp, clip = (native.OP_AFFINE, 0, 1.0, -273.15), (native.OP_MUL, 0, 9.80665, 0.0), (native.OP_COPY, 0, 0.0, 0.0), (native.OP_CLIP, 0, 200.0, 320.0)
    for n_var in (1, 3):
        n_lev = 137 * n_var
        programs = {
            "3 runs (affine | mul | copy), boundaries off the vector grid": [[aff] * 50 + [mul] * 50 + [cp] * (n_lev - 100)] if n_var == 1 else [[aff] * 137 + [mul] * 137 + [cp] * 137],
            "2 stages x 3 runs": [[cp] * 45 + [mul] * 45 + [cp] * (n_lev - 90), [aff] * 45 + [cp] * 45 + [clip] * (n_lev - 90)] if n_var == 1 else
                                 [[cp] * 137 + [mul] * 137 + [cp] * 137, [aff] * 137 + [cp] * 137 + [clip] * 137],
            "uniform (one affine over all levels: the by-value kernel of round 3)": [[aff] * n_lev],
        }
        for dt in (torch.float32, torch.float64):
            x = Stack.empty(n_pts, n_lev, dt, dev, COLUMNS, zero=True)
            x.data[:, :n_lev].normal_(270.0, 15.0)
            y = x.new_like()
            nbytes = 2 * n_pts * n_lev * x.data.element_size()
            for name, stages in programs.items():
                prog = native.level_program(stages, dev)
                kw = dict(n_pts=n_pts, n_lev=n_lev, x_pitch=x.pitch, y_pitch=y.pitch, layout=COLUMNS, prog=prog, n_stage=len(stages))
                ms = timeit(lambda: native.pointwise_stack(x.data, y.data, **kw))
                print(f"{n_lev:4d} levels {str(dt).split('.')[-1]:8s} {name:70s} {ms:8.4f} ms  frac {nbytes / ms / 1e6 / 8000:.3f}")
            del x, y
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
