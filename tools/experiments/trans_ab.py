#!/usr/bin/env python3
"""Interleaved A/B of libatx builds on the per-point kernels that evaluate library functions (float64 log / exp, snow_cover's tanh):

    python tools/experiments/trans_ab.py base=anemoi-transform_amd/lib/libatx.so u2=anemoi-transform_amd/lib/variants/libatx_u2.so ...

137 levels of O1280; every (library, case) pair is timed in every round, rounds interleaved; median ms and the fraction of 8 TB/s on
algorithmic bytes.  The first library's outputs are the reference the others are compared with (max ulp distance, NaN positions)."""
from __future__ import annotations

import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402

PEAK = 8.0e12


def main():
    graft.load_package()
    from anemoi_transform_amd import native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    rounds = int(os.environ.get("ATX_AB_ROUNDS", "5"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    libs = {}
    for spec in sys.argv[1:]:
        name, path = spec.split("=", 1)
        h = ctypes.CDLL(path if os.path.isabs(path) else os.path.join(ROOT, path))
        for fn, (restype, argtypes) in native.SIGNATURES.items():
            getattr(h, fn).restype = restype
            getattr(h, fn).argtypes = argtypes
        libs[name] = h
    grid = lookup("o1280")
    n, L = len(grid["latitudes"]), 137
    cases = {}
    for tdt, B, tag in ((torch.float64, 8, "f64"), (torch.float32, 4, "f32")):
        x = bench.synth_stack(grid, L, tdt, dev, 0, COLUMNS)  # K-like magnitudes: positive, fine for log
        y = x.new_like()
        kw = dict(n_pts=n, n_lev=L, x_pitch=x.pitch, y_pitch=y.pitch, layout=COLUMNS)
        nb = n * L * B

        def prog(*ops):
            return native.level_program([[(op, 0, a, b)] * L for op, a, b in ops], dev)

        lnx = x.new_like()
        lnx.data.copy_(torch.log(x.data.clamp_min(1.0)))
        programs = {"affine": (x, prog((native.OP_AFFINE, 1.0, -273.15)), 1), "log": (x, prog((native.OP_LOG, 0.0, 0.0)), 1),
                    "exp": (lnx, prog((native.OP_EXP, 0.0, 0.0)), 1), "log then exp": (x, prog((native.OP_LOG, 0.0, 0.0), (native.OP_EXP, 0.0, 0.0)), 2)}
        for name, (src, p, ns) in programs.items():
            cases[f"pointwise {name} {tag}"] = (lambda src=src, p=p, ns=ns, y=y, kw=kw: native.pointwise_stack(src.data, y.data, prog=p, n_stage=ns, **kw), 2 * nb, y)
        if tag == "f64" or os.environ.get("ATX_AB_F32_COMBINE"):
            sd, rsn, z = x.new_like(), x.new_like(), x.new_like()
            rsn.data[:, :L] = (100.0 + 300.0 * torch.rand(n, L, device=dev)).to(tdt)
            u = (torch.arange(n, device=dev, dtype=torch.float64) / n).unsqueeze(1).expand(n, L)
            sd_regions = torch.where(u < 0.55, torch.zeros_like(u), torch.where(u < 0.9, 0.05 + u, 1e-4 * u)).to(tdt)
            del u
            sd_thin = x.new_like()
            sd_thin.data.fill_(1e-5)
            sd.data[:, :L] = sd_regions
            del sd_regions
            ckw = dict(n_pts=n, n_lev=L, pitch=x.pitch, layout=COLUMNS)
            cases[f"snow_cover regions {tag}"] = (lambda sd=sd, rsn=rsn, z=z, ckw=ckw: native.combine_stack(native.COMB_SNOW_COVER, [sd.data, rsn.data], [z.data], **ckw), 3 * nb, z)
            cases[f"snow_cover thin cover everywhere {tag}"] = (lambda sd=sd_thin, rsn=rsn, z=z, ckw=ckw: native.combine_stack(native.COMB_SNOW_COVER, [sd.data, rsn.data], [z.data], **ckw), 3 * nb, z)
            ang = x.new_like()
            ang.data[:, :L] = ((torch.rand(n, L, device=dev, dtype=torch.float64) * 2.0 - 1.0) * 6.283185307179586).to(tdt)  # wave directions / phases in [-2 pi, 2 pi]
            z2 = x.new_like()
            cases[f"cos_sin {tag}"] = (lambda a=ang, z=z, z2=z2, ckw=ckw: native.combine_stack(native.COMB_COS_SIN, [a.data], [z.data, z2.data], **ckw), 3 * nb, z)
            cases[f"atan2 (cos, sin -> direction) {tag}"] = (lambda a=x, b=rsn, z=z, ckw=ckw: native.combine_stack(native.COMB_ATAN2, [a.data, b.data], [z.data], **ckw), 3 * nb, z)
            cases[f"atan2 in degrees, wrapped to [0, 360) {tag}"] = (lambda a=x, b=rsn, z=z, ckw=ckw: native.combine_stack(native.COMB_ATAN2, [a.data, b.data], [z.data], flags=native.COMB_DEGREES, **ckw), 3 * nb, z)
            cases[f"difference {tag}"] = (lambda a=x, b=rsn, z=z, ckw=ckw: native.combine_stack(native.COMB_SUB, [a.data, b.data], [z.data], **ckw), 3 * nb, z)

    def time_once(fn, reps=8):
        for _ in range(2):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs]))

    times = {c: {l: [] for l in libs} for c in cases}
    agree = {}
    for r in range(rounds):
        for c, (fn, nbytes, out) in cases.items():
            ref = None
            for l, h in libs.items():
                native._lib = h
                times[c][l].append(time_once(fn))
                if r == 0:  # outputs of every build against the first one's
                    got = out.data.clone()
                    if ref is None:
                        ref = got
                    else:
                        same_nan = bool(torch.equal(torch.isnan(got), torch.isnan(ref)))
                        it = torch.int64 if got.dtype == torch.float64 else torch.int32
                        fin = ~torch.isnan(ref)
                        d = (got.view(it)[fin] - ref.view(it)[fin]).abs().max().item() if fin.any() else 0
                        agree[f"{c} | {l}"] = {"max_ulp_distance_to_first": int(d), "same_nan_positions": same_nan}
    res = {}
    for c, (fn, nbytes, out) in cases.items():
        for l in libs:
            ms = float(np.median(times[c][l]))
            res[f"{c} | {l}"] = {"ms": ms, "min_ms": float(np.min(times[c][l])), "frac_of_8TBs": nbytes / (ms * 1e-3) / PEAK, **agree.get(f"{c} | {l}", {})}
            print(f"{c:46s} {l:12s} {ms:8.4f} ms (min {np.min(times[c][l]):.4f})  frac {nbytes / (ms * 1e-3) / PEAK:.3f}  {agree.get(f'{c} | {l}', '')}", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
