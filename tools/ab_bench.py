#!/usr/bin/env python3
"""Interleaved A/B timing of libatx builds on the bench workload (one process, one GPU).

    python tools/ab_bench.py --libs base=anemoi-transform_amd/lib/libatx.so nt=anemoi-transform_amd/lib/variants/libatx_nt_store.so \
        --tiles 0 16 24 --rounds 7 --cases k4f32 k1f32 k4f64

Every (library, tile) pair is timed in every round, rounds interleaved
(cdna_hip_programming.md §5.4 rule 24); reports median / min ms per launch and the
roofline fraction on algorithmic bytes.
"""

from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True, help="name=path pairs")
    ap.add_argument("--tiles", nargs="+", type=int, default=[0])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--inner", type=int, default=10)
    ap.add_argument("--cases", nargs="+", default=["k4f32"])
    ap.add_argument("--levels", type=int, default=137)
    ap.add_argument("--src-grid", default="o1280")
    ap.add_argument("--tgt-grid", default="0.25")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    graft.load_package()
    from anemoi_transform_amd import native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.interp import knn_inverse_distance
    from anemoi_transform_amd.stack import COLUMNS, Stack

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    libs = {}
    for spec in args.libs:
        name, path = spec.split("=", 1)
        h = ctypes.CDLL(os.path.join(ROOT, path) if not os.path.isabs(path) else path)
        for fn, (restype, argtypes) in native.SIGNATURES.items():
            getattr(h, fn).restype = restype
            getattr(h, fn).argtypes = argtypes
        libs[name] = h

    src_grid, tgt_grid = lookup(args.src_grid), lookup(args.tgt_grid)
    n_src, n_tgt = len(src_grid["latitudes"]), len(tgt_grid["latitudes"])
    idx64, w64 = knn_inverse_distance(src_grid, tgt_grid, k=4)
    stream = torch.cuda.current_stream().cuda_stream
    results = {}
    for case in args.cases:
        k = int(case[1])
        f64 = "f64" in case
        tdt, npdt, isz = (torch.float64, np.float64, 8) if f64 else (torch.float32, np.float32, 4)
        src = bench.synth_stack(src_grid, args.levels, tdt, dev, 0, COLUMNS)
        out = Stack.empty(n_tgt, args.levels, tdt, dev, COLUMNS)
        idx = torch.from_numpy(np.ascontiguousarray(idx64[:, :k]).astype(np.int32)).to(dev)
        w = torch.from_numpy(w64.astype(npdt)).to(dev) if k > 1 else None
        alg = bench.algorithmic_bytes(args.levels, isz, int(np.unique(idx64[:, :k]).size), n_tgt, k)

        # "...e": with the 2-stage epilogue of config 5 (x * g, then x - 273.15)
        prog = native.level_program([[(native.OP_MUL, 0, 9.80665, 0.0)] * args.levels, [(native.OP_AFFINE, 0, 1.0, -273.15)] * args.levels],
                                    dev) if "e" in case[2:] else None
        f64 = "f64" in case

        vec_p, host_p = native._program_companions(prog, tdt)

        def run(h, tile):
            h.atx_set_tuning(tile)
            rc = h.atx_regrid_ell(src.data.data_ptr(), out.data.data_ptr(), idx.data_ptr(), None if w is None else w.data_ptr(),
                                  n_src, n_tgt, k, args.levels, src.pitch, out.pitch, 1 if f64 else 0, 0, 0,
                                  None if prog is None else prog.data_ptr(), vec_p, host_p, 0 if prog is None else 2, None, stream)
            assert rc == 0, h.atx_last_error()

        combos = [(name, tile) for name in libs for tile in args.tiles]
        times = {c: [] for c in combos}
        ref_out = None
        for c in combos:  # warm-up + every variant must produce the same bits
            out.data.fill_(float("nan"))
            run(libs[c[0]], c[1])
            torch.cuda.synchronize()
            if ref_out is None:
                ref_out = out.data.clone()
            else:
                same = torch.equal(out.data.view(torch.int32 if not f64 else torch.int64), ref_out.view(torch.int32 if not f64 else torch.int64))
                assert same, f"variant {c} differs from {combos[0]}"
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for c in combos:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.inner):
                    run(libs[c[0]], c[1])
                b.record()
                torch.cuda.synchronize()
                times[c].append(a.elapsed_time(b) / args.inner)
        for c in combos:
            med, mn = float(np.median(times[c])), float(np.min(times[c]))
            results[f"{case}/{c[0]}/tile{c[1]}"] = {"median_ms": med, "min_ms": mn, "frac_median": alg / (med * 1e-3) / 8e12,
                                                    "frac_min": alg / (mn * 1e-3) / 8e12}
            print(f"{case:6s} {c[0]:14s} tile={c[1]:3d}  median {med:.4f} ms  min {mn:.4f} ms  frac {alg / (med * 1e-3) / 8e12:.4f}", flush=True)
        del src, out
        torch.cuda.empty_cache()
    if args.out:
        json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
