#!/usr/bin/env python3
"""atx_reduce_stack (min+max in one pass, min, NaN count) over 137 levels of O1280 and over one field, for an A/B of library builds
(ATX_LIBRARY=...; -DATX_RED_GRID=<workgroup cap>: the partials of the two-level finish scale with it)."""
from __future__ import annotations

import os
import sys

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
    n_pts = 6599680
    print(f"# library: {native.lib_path()}")
    for dt in (torch.float32, torch.float64):
        for n_lev in (137, 1):
            x = Stack.empty(n_pts, n_lev, dt, dev, COLUMNS, zero=True)
            x.data[:, :n_lev].normal_(270.0, 15.0)
            kw = dict(n_pts=n_pts, n_lev=n_lev, pitch=x.pitch, layout=COLUMNS)
            want = (float(x.data[:, :n_lev].min()), float(x.data[:, :n_lev].max()))
            assert native.reduce_stack(x.data, native.RED_MINMAX, **kw) == want
            nbytes = n_pts * n_lev * x.data.element_size()
            for name, red in (("min+max", native.RED_MINMAX), ("nan count", native.RED_NANCOUNT)):
                ms = timeit(lambda: native.reduce_stack(x.data, red, **kw), n=30)
                print(f"{str(dt).split('.')[-1]:8s} {n_lev:4d} levels {name:10s} {ms * 1e3:9.1f} us  frac {nbytes / ms / 1e6 / 8000:.3f}")
            del x
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
