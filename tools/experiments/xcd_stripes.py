#!/usr/bin/env python3
"""Round-4 experiment: how the workgroups of the direct gather kernel are dealt to the XCDs (ATX_DIRECT_STRIPE builds), targets in NATURAL
order against the column-block order of round 3.  tools/experiments/tile_orders.py showed that only FULL-HEIGHT column blocks help long
rows — bands of limited height and 2-D tiles do not — i.e. the gain is not vertical reuse in L2 but balance: with one contiguous range of
targets per XCD (xcd_tile) the XCDs that hold polar rows (every source column shared by many targets) finish early and the equatorial
ones carry the launch; striping the ranges over the XCDs balances the latitudes without permuting anything.

    python tools/experiments/xcd_stripes.py --libs head=anemoi-transform_amd/lib/libatx.so s64=anemoi-transform_amd/lib/variants/libatx_stripe64.so ...
"""

from __future__ import annotations

import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--inner", type=int, default=10)
    ap.add_argument("--ks", nargs="+", type=int, default=[16, 8, 4])
    ap.add_argument("--dtypes", nargs="+", default=["f32", "f64"])
    ap.add_argument("--block-points", nargs="+", type=int, default=[0], help="widths of the column blocks (0: the package's default)")
    args = ap.parse_args()
    graft.load_package()
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.gather import column_block_order
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS, Stack

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    native.load()
    libs = {}
    for spec in args.libs:
        name, path = spec.split("=", 1)
        h = ctypes.CDLL(path if os.path.isabs(path) else os.path.join(ROOT, path))
        for fn in ("atx_regrid_ell", "atx_regrid_ell_ordered"):
            restype, argtypes = native.SIGNATURES[fn]
            getattr(h, fn).restype, getattr(h, fn).argtypes = restype, argtypes
        libs[name] = h
    L = 137
    src_grid, tgt_grid = lookup("o1280"), lookup("0.25")
    n_src, n_tgt = len(src_grid["latitudes"]), len(tgt_grid["latitudes"])
    idx16, w16 = interp.knn_inverse_distance(src_grid, tgt_grid, k=16, device=True, ties="index")
    orders = {("column blocks" if not bp else f"blocks of {bp}"): column_block_order(tgt_grid["latitudes"], tgt_grid["longitudes"], bp or None)
              for bp in args.block_points}
    rows_dev = {name: torch.from_numpy(o.astype(np.int32)).to(dev) for name, o in orders.items()}
    stream = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731

    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.inner):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.inner

    for tag in args.dtypes:
        tdt, npdt, B = (torch.float32, np.float32, 4) if tag == "f32" else (torch.float64, np.float64, 8)
        x = bench.synth_stack(src_grid, L, tdt, dev, 0, COLUMNS)
        out = Stack.empty(n_tgt, L, tdt, dev, COLUMNS)
        for k in args.ks:
            alg = bench.algorithmic_bytes(L, B, int(np.unique(idx16[:, :k]).size), n_tgt, k)
            wk = w16[:, :k] / w16[:, :k].sum(axis=1, keepdims=True)
            tables = {}
            for oname, perm in [("natural", None)] + list(orders.items()):
                i = idx16[:, :k] if perm is None else idx16[perm, :k]
                w = wk if perm is None else wk[perm]
                tables[oname] = (torch.from_numpy(np.ascontiguousarray(i).astype(np.int32)).to(dev), torch.from_numpy(np.ascontiguousarray(w).astype(npdt)).to(dev))
            results: dict = {}

            def call(h, oname):
                ik, wd = tables[oname]
                tail = (n_src, n_tgt, k, L, x.pitch, out.pitch, native.dtype_code(tdt), COLUMNS, 0, None, None, None, 0, None, stream())
                if oname == "natural":
                    rc = h.atx_regrid_ell(x.data.data_ptr(), out.data.data_ptr(), ik.data_ptr(), wd.data_ptr(), *tail)
                else:
                    srcs, outs = (ctypes.c_void_p * 1)(x.data.data_ptr()), (ctypes.c_void_p * 1)(out.data.data_ptr())
                    rc = h.atx_regrid_ell_ordered(srcs, outs, 1, ik.data_ptr(), wd.data_ptr(), rows_dev[oname].data_ptr(), *tail)
                assert rc == 0, rc

            for rnd in range(args.rounds + 1):
                for oname in tables:
                    for lname, h in libs.items():
                        ms = timed(lambda: call(h, oname))
                        if rnd:
                            results.setdefault((oname, lname), []).append(ms)
            for (oname, lname), v in results.items():
                med = float(np.median(v))
                print(f"{tag} k={k:2d} {oname:14s} {lname:8s} median {med:7.4f} ms  min {min(v):7.4f}  frac {alg / med / 1e9 / 8:.3f}", flush=True)
        del x, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
