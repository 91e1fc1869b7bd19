#!/usr/bin/env python3
"""Layout conversion both ways, float32 and float64, 137 levels of O1280 (and a thin 13-level stack): median HIP-event time, fraction of the
HBM peak on 2 * N * L * B bytes, bit-equality with torch's own transpose.  `ATX_LIBRARY=<variant>` selects another build for an A/B."""
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


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n_pts = 6599680
    print(f"# library: {native.lib_path()}")
    for n_lev in (137, 13):
        for dt in (torch.float32, torch.float64):
            x = Stack.empty(n_pts, n_lev, dt, dev, COLUMNS, zero=True)
            x.data[:, :n_lev] = torch.randn(n_pts, n_lev, dtype=dt, device=dev)
            f = Stack.empty(n_pts, n_lev, dt, dev, FIELDS)
            y = Stack.empty(n_pts, n_lev, dt, dev, COLUMNS, zero=True)
            kw_cf = dict(n_pts=n_pts, n_lev=n_lev, src_pitch=x.pitch, dst_pitch=f.pitch, src_layout=COLUMNS, dst_layout=FIELDS)
            kw_fc = dict(n_pts=n_pts, n_lev=n_lev, src_pitch=f.pitch, dst_pitch=y.pitch, src_layout=FIELDS, dst_layout=COLUMNS)
            native.relayout(x.data, f.data, **kw_cf)
            native.relayout(f.data, y.data, **kw_fc)
            ok = torch.equal(f.data[:, :n_pts], x.data[:, :n_lev].T) and torch.equal(y.data[:, :n_lev], x.data[:, :n_lev])
            nbytes = 2 * n_pts * n_lev * x.data.element_size()
            for name, fn in (("columns->fields", lambda: native.relayout(x.data, f.data, **kw_cf)), ("fields->columns", lambda: native.relayout(f.data, y.data, **kw_fc))):
                ms = timeit(fn)
                print(f"{n_lev:4d} levels {str(dt).split('.')[-1]:8s} {name:16s} {ms:8.4f} ms  {nbytes / ms / 1e6:8.1f} GB/s  frac {nbytes / ms / 1e6 / 8000:.3f}  bits {'ok' if ok else 'DIFFER'}")
            del x, f, y
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
