#!/usr/bin/env python3
"""Every operator of atx_combine_stack over 137 levels of O1280 (column stacks): time and fraction of 8 TB/s on (n_in + n_out) x stack bytes."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402
from per_level_programs import launches  # noqa: E402


def main():
    graft.load_package()
    from anemoi_transform_amd import native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    src = lookup("o1280")
    n, L = len(src["latitudes"]), 137
    print("library:", native.lib_path(), flush=True)
    for tdt, B, tag in ((torch.float32, 4, "f32"), (torch.float64, 8, "f64")):
        t = bench.synth_stack(src, L, tdt, dev, 0, COLUMNS)  # temperatures ~ 200 .. 320
        u, v = t.new_like(), t.new_like()
        u.data.copy_(t.data).sub_(270.0).mul_(0.4)  # winds ~ -28 .. 20
        v.data.copy_(t.data).sub_(255.0).mul_(-0.3)
        ang = t.new_like()
        ang.data.copy_(t.data).sub_(200.0).mul_(0.05)  # radians in [0, 6]
        deg = t.new_like()
        deg.data.copy_(t.data).sub_(200.0).mul_(2.9)  # degrees in [0, 350]
        o1, o2 = t.new_like(), t.new_like()
        lev = torch.linspace(1.0, 1000.0, L, dtype=torch.float64, device=dev) * 100.0
        kw = dict(n_pts=n, n_lev=L, pitch=t.pitch, layout=COLUMNS)
        cases = [
            ("snow_depth_m (sd, rsn)", native.COMB_SNOW_DEPTH_M, [u, t], [o1], {}),
            ("atan2 (sin, cos) -> direction", native.COMB_ATAN2, [u, v], [o1], {}),
            ("cos_sin (radians)", native.COMB_COS_SIN, [ang], [o1, o2], {}),
            ("cos_sin (degrees)", native.COMB_COS_SIN, [deg], [o1, o2], {"flags": native.COMB_DEGREES}),
            ("w_to_wz (w, t, q)", native.COMB_W_TO_WZ, [u, t, ang], [o1], {"level_param": lev}),
            ("wz_to_w (wz, t, q)", native.COMB_WZ_TO_W, [u, t, ang], [o1], {"level_param": lev}),
            ("sum of 3", native.COMB_SUM, [u, v, t], [o1], {}),
            ("difference", native.COMB_SUB, [u, v], [o1], {}),
            ("xy_to_polar (u, v) -> speed, direction", native.COMB_XY_TO_POLAR, [u, v], [o1, o2], {}),
            ("polar_to_xy (speed, direction) -> u, v", native.COMB_POLAR_TO_XY, [t, deg], [o1, o2], {}),
        ]
        for name, op, ins, outs, extra in cases:
            ms = launches(lambda: native.combine_stack(op, [a.data for a in ins], [a.data for a in outs], **extra, **kw))
            print(f"{tag} {name:42s} {ms:7.3f} ms  {(len(ins) + len(outs)) * n * L * B / (ms * 1e-3) / 8e12:.3f}", flush=True)
        del t, u, v, ang, deg, o1, o2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
