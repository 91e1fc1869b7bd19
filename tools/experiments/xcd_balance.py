#!/usr/bin/env python3
"""Round-4 experiment: contiguous workgroup ranges of equal COST per XCD (atx_set_xcd_targets) against the default ranges of equal length,
targets in natural order, O1280 -> 0.25 degree, 137 levels.  Cost of a target = its output row + the source columns it is the FIRST to
reference (what an XCD has to fetch from HBM when it walks the targets in order), optionally + a share per access that hits.

    python tools/experiments/xcd_balance.py

Needs the prototype of tools/experiments/xcd_balance.patch applied to csrc/ (a by-value table of workgroup ranges in the direct kernel and
the hook atx_set_xcd_targets): NOT adopted — ranges of equal HBM cost are 6-37 % slower than ranges of equal length (the polar XCDs, with
up to 23 % of the targets, become the tail: their accesses hit but are not free), and the best weighting of hits found gains 2-6 % at
k = 16 / 8 and nothing at k <= 4, less than the column-block order already in use (profiles/r04_xcd_balance_experiment.log).
"""

from __future__ import annotations

import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def boundaries(idx: np.ndarray, n_src: int, hit_cost: float) -> np.ndarray:
    """9 target indices cutting the targets (in table order) into 8 ranges of equal cost."""
    n_tgt, k = idx.shape
    first = np.full(n_src, n_tgt, dtype=np.int64)
    np.minimum.at(first, idx.reshape(-1), np.repeat(np.arange(n_tgt), k))
    fetched = np.bincount(first[first < n_tgt], minlength=n_tgt).astype(np.float64)  # columns first referenced at each target
    cost = 1.0 + fetched + hit_cost * (k - fetched)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = np.searchsorted(cum, np.linspace(0.0, cum[-1], 9)[1:-1])
    return np.concatenate([[0], cuts, [n_tgt]]).astype(np.int64)


def main():
    graft.load_package()
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS, Stack

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    lib = native.load()
    lib.atx_set_xcd_targets.restype = ctypes.c_int
    lib.atx_set_xcd_targets.argtypes = [ctypes.c_void_p]
    L = 137
    src_grid, tgt_grid = lookup("o1280"), lookup("0.25")
    n_src, n_tgt = len(src_grid["latitudes"]), len(tgt_grid["latitudes"])
    idx16, w16 = interp.knn_inverse_distance(src_grid, tgt_grid, k=16, device=True, ties="index")

    def timed(fn, inner=10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / inner

    for tag, tdt, npdt, B in (("f64", torch.float64, np.float64, 8), ("f32", torch.float32, np.float32, 4)):
        x = bench.synth_stack(src_grid, L, tdt, dev, 0, COLUMNS)
        out = Stack.empty(n_tgt, L, tdt, dev, COLUMNS)
        ref = Stack.empty(n_tgt, L, tdt, dev, COLUMNS)
        for k in (4, 1, 16, 8):
            idx = np.ascontiguousarray(idx16[:, :k])
            alg = bench.algorithmic_bytes(L, B, int(np.unique(idx).size), n_tgt, k)
            wk = w16[:, :k] / w16[:, :k].sum(axis=1, keepdims=True)
            ik = torch.from_numpy(idx.astype(np.int32)).to(dev)
            wd = None if k == 1 else torch.from_numpy(np.ascontiguousarray(wk).astype(npdt)).to(dev)
            kw = dict(n_src=n_src, n_tgt=n_tgt, k=k, n_lev=L, src_pitch=x.pitch, out_pitch=out.pitch, layout=COLUMNS)
            settings = {"equal length": None}
            for hit in (0.0, 0.05, 0.1, 0.2, 0.4):
                settings[f"equal cost, hit {hit}"] = boundaries(idx, n_src, hit)
            lib.atx_set_xcd_targets(None)
            native.regrid_ell(x.data, ref.data, ik, wd, **kw)
            results = {}
            for rnd in range(6):
                for name, cuts in settings.items():
                    lib.atx_set_xcd_targets(None if cuts is None else cuts.ctypes.data_as(ctypes.c_void_p))
                    ms = timed(lambda: native.regrid_ell(x.data, out.data, ik, wd, **kw))
                    if rnd == 0:
                        assert torch.equal(out.data.view(torch.uint8), ref.data.view(torch.uint8)), name
                    else:
                        results.setdefault(name, []).append(ms)
            lib.atx_set_xcd_targets(None)
            for name, v in results.items():
                cuts = settings[name]
                shares = "" if cuts is None else "  targets per XCD (%): " + " ".join(f"{100 * (b - a) / n_tgt:.1f}" for a, b in zip(cuts, cuts[1:]))
                med = float(np.median(v))
                print(f"{tag} k={k:2d} {name:22s} median {med:7.4f} ms  min {min(v):7.4f}  frac {alg / med / 1e9 / 8:.3f}{shares}", flush=True)
        del x, out, ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
