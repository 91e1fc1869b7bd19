#!/usr/bin/env python3
"""Interleaved A/B of `atx_pointwise_stack` across builds of libatx in ONE process on ONE GPU (round 3: the f64 per-point
"regression" the round-2 judge traced to commit 702e31f).

    bash tools/build_variant.sh pre702e31f --rev 702e31f~1
    bash tools/build_variant.sh table0 -DATX_PW_TABLE_STAGES=0          # round 2's dispatch rule, today's sources
    bash tools/build_variant.sh nocap  -DATX_PW_TABLE_STAGES=0 -DATX_MAX_GRID=2147483647
    python tools/experiments/pointwise_ab.py --libs head=anemoi-transform_amd/lib/libatx.so \
        pre=anemoi-transform_amd/lib/variants/libatx_pre702e31f.so:old table0=... nocap=...

A `:old` suffix marks a build with the 13-argument entry point (before `host_prog` was added).  Cases: one- and two-stage affine
programs, f32 / f64, out of place / in place, with and without the point mask; the library's fixed `atx_stream_copy` of the same
bytes is timed in every round as the yardstick."""

from __future__ import annotations

import argparse
import ctypes
import os
import sys
from ctypes import c_int, c_int32, c_int64, c_void_p

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--inner", type=int, default=10)
    ap.add_argument("--levels", type=int, default=137)
    ap.add_argument("--dtypes", nargs="+", default=["f64", "f32"])
    args = ap.parse_args()
    graft.load_package()
    from anemoi_transform_amd import native
    from anemoi_transform_amd.stack import COLUMNS, column_pitch

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    native.load()
    libs = {}
    for spec in args.libs:
        name, path = spec.split("=", 1)
        old = path.endswith(":old")
        path = path[:-4] if old else path
        h = ctypes.CDLL(path if os.path.isabs(path) else os.path.join(ROOT, path))
        sig = [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]
        sig += ([] if old else [c_void_p]) + [c_int32, c_void_p, c_void_p]
        h.atx_pointwise_stack.restype = c_int
        h.atx_pointwise_stack.argtypes = sig
        libs[name] = (h, old)

    L, n = args.levels, 6_599_680
    stream = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731

    def timed(fn, inner):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / inner

    for tag in args.dtypes:
        tdt, B = (torch.float64, 8) if tag == "f64" else (torch.float32, 4)
        pitch = column_pitch(L, tdt)
        x = torch.rand(n * pitch, dtype=tdt, device=dev).view(n, pitch)
        y = torch.empty_like(x)
        pm = (torch.rand(n + 8, device=dev) < 0.3).to(torch.uint8)
        aff = (native.OP_AFFINE, 0, 1.0, -273.15)
        progs = {
            "1 stage": (native.level_program([[aff] * L], dev), 1, None),
            "2 stages": (native.level_program([[(native.OP_MUL, 0, 9.80665, 0.0)] * L, [aff] * L], dev), 2, None),
            "3 stages": (native.level_program([[(native.OP_MUL, 0, 9.80665, 0.0)] * L, [aff] * L, [(native.OP_CLIP, 0, -1e30, 1e30)] * L], dev), 3, None),
            "apply_mask": (native.level_program([[(native.OP_COPY, 1, 0.0, 0.0)] * L], dev), 1, pm),
        }
        alg = 2 * n * L * B
        cases = []
        for pname, (prog, n_stage, mask) in progs.items():
            for place in ("out", "in"):
                cases.append((f"{tag} {pname:10s} {place:3s}", prog, n_stage, mask, place))
        results: dict = {}

        def call(h, old, prog, n_stage, mask, place):
            vec, host = native._program_companions(prog, tdt)
            src = y if place == "in" else x
            a = [src.data_ptr(), y.data_ptr(), n, L, pitch, pitch, native.dtype_code(tdt), COLUMNS, prog.data_ptr(), vec]
            a += ([] if old else [host]) + [n_stage, mask.data_ptr() if mask is not None else None, stream()]
            rc = h.atx_pointwise_stack(*a)
            assert rc == 0, rc

        for rnd in range(args.rounds + 1):  # round 0 warms up
            ms = timed(lambda: native.stream_copy(x, y), args.inner)
            if rnd:
                results.setdefault((f"{tag} atx_stream_copy (yardstick)", "-"), []).append(ms)
            for cname, prog, n_stage, mask, place in cases:
                for lname, (h, old) in libs.items():
                    ms = timed(lambda: call(h, old, prog, n_stage, mask, place), args.inner)
                    if rnd:
                        results.setdefault((cname, lname), []).append(ms)
        for (cname, lname), v in results.items():
            med = float(np.median(v))
            print(f"{cname:34s} {lname:10s} median {med:7.4f} ms  min {min(v):7.4f}  frac {alg / med / 1e9 / 8:.3f}", flush=True)
        del x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
