#!/usr/bin/env python3
"""Interleaved A/B of `atx_combine_stack` across builds of libatx in ONE process (137 levels of O1280).

    python tools/experiments/combine_ab.py --libs head=anemoi-transform_amd/lib/libatx.so u1=anemoi-transform_amd/lib/variants/libatx_c_u1.so ...

Cases: difference (2 -> 1), cos+sin (1 -> 2), snow_cover (2 -> 1) on a field with snow in regions and with a thin cover everywhere
(tanh on every element), w_to_wz (3 -> 1); float32 and float64.  atx_stream_copy of one stack is the yardstick."""

from __future__ import annotations

import argparse
import ctypes
import os
import sys
from ctypes import c_void_p

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", required=True)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--inner", type=int, default=8)
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
        h = ctypes.CDLL(path if os.path.isabs(path) else os.path.join(ROOT, path))
        restype, argtypes = native.SIGNATURES["atx_combine_stack"]
        h.atx_combine_stack.restype, h.atx_combine_stack.argtypes = restype, argtypes
        libs[name] = h
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
        new = lambda: torch.zeros(n, pitch, dtype=tdt, device=dev)  # noqa: E731
        x, y, z, q = new(), new(), new(), new()
        x[:, :L] = (280.0 + 30.0 * torch.randn(n, L, device=dev)).to(tdt)
        q[:, :L] = (0.01 * torch.rand(n, L, device=dev)).to(tdt)
        u = (torch.arange(n, device=dev, dtype=torch.float64) / n).unsqueeze(1).expand(n, L)
        sd_regions, sd_thin, rsn = new(), new(), new()
        sd_regions[:, :L] = torch.where(u < 0.55, torch.zeros_like(u), torch.where(u < 0.9, 0.05 + u, 1e-4 * u)).to(tdt)
        sd_thin[:, :L] = 1e-5
        rsn[:, :L] = (100.0 + 300.0 * torch.rand(n, L, device=dev)).to(tdt)
        del u
        plev = torch.linspace(1.0, 1000.0, L, dtype=torch.float64, device=dev)
        stack = n * L * B
        cases = {
            "difference 2->1": (native.COMB_SUB, [x, rsn], [y], None, 3 * stack),
            "cos_sin 1->2": (native.COMB_COS_SIN, [q], [y, z], None, 3 * stack),
            "snow_cover regions": (native.COMB_SNOW_COVER, [sd_regions, rsn], [y], None, 3 * stack),
            "snow_cover thin": (native.COMB_SNOW_COVER, [sd_thin, rsn], [y], None, 3 * stack),
            "w_to_wz 3->1": (native.COMB_W_TO_WZ, [x, rsn, q], [y], plev, 4 * stack),
        }
        results: dict = {}

        def call(h, op, ins, outs, lp):
            i = (c_void_p * len(ins))(*[t.data_ptr() for t in ins])
            o = (c_void_p * len(outs))(*[t.data_ptr() for t in outs])
            rc = h.atx_combine_stack(op, i, len(ins), o, len(outs), n, L, pitch, native.dtype_code(tdt), COLUMNS,
                                     lp.data_ptr() if lp is not None else None, 0, stream())
            assert rc == 0, rc

        for rnd in range(args.rounds + 1):
            ms = timed(lambda: native.stream_copy(x, y), args.inner)
            if rnd:
                results.setdefault((f"{tag} atx_stream_copy of one stack", "-", 2 * n * pitch * B), []).append(ms)
            for cname, (op, ins, outs, lp, alg) in cases.items():
                for lname, h in libs.items():
                    ms = timed(lambda: call(h, op, ins, outs, lp), args.inner)
                    if rnd:
                        results.setdefault((f"{tag} {cname}", lname, alg), []).append(ms)
        for (cname, lname, alg), v in results.items():
            med = float(np.median(v))
            print(f"{cname:34s} {lname:10s} median {med:7.4f} ms  min {min(v):7.4f}  frac {alg / med / 1e9 / 8:.3f}", flush=True)
        del x, y, z, q, sd_regions, sd_thin, rsn
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
