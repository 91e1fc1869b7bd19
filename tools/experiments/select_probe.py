#!/usr/bin/env python3
"""atx_select_levels on a 137-level O1280 column stack: every other level, the first 68, one level — median HIP-event time and bit-equality."""
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
from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack  # noqa: E402
from relayout_probe import timeit  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n_pts, n_lev = 6599680, 137
    for layout in (COLUMNS, FIELDS):
        for dt in (torch.float32, torch.float64):
            x = Stack.empty(n_pts, n_lev, dt, dev, layout, zero=True)
            if layout == COLUMNS:
                x.data[:, :n_lev] = torch.randn(n_pts, n_lev, dtype=dt, device=dev)
            else:
                x.data[:, :n_pts] = torch.randn(n_lev, n_pts, dtype=dt, device=dev)
            for name, lm in (("every other level (68)", list(range(0, 136, 2))), ("first 68 levels", list(range(68))), ("all 137 reversed", list(range(136, -1, -1))), ("1 level", [77])):
                out = Stack.empty(n_pts, len(lm), dt, dev, layout, zero=True)
                fn = lambda: native.select_levels(x.data, out.data, lm, n_pts=n_pts, n_src_lev=n_lev, src_pitch=x.pitch, dst_pitch=out.pitch, layout=layout)  # noqa: E731
                fn()
                want = x.data[:, lm] if layout == COLUMNS else x.data[lm, :n_pts]
                got = out.data[:, :len(lm)] if layout == COLUMNS else out.data[:, :n_pts]
                ok = torch.equal(got, want)
                ms = timeit(fn)
                nbytes = 2 * n_pts * len(lm) * x.data.element_size()
                print(f"{'columns' if layout == COLUMNS else 'fields':8s} {str(dt).split('.')[-1]:8s} {name:24s} {ms:8.4f} ms  frac {nbytes / ms / 1e6 / 8000:.3f} (algorithmic)  bits {'ok' if ok else 'DIFFER'}")
                del out
            del x
            torch.cuda.empty_cache()


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
