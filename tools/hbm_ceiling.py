#!/usr/bin/env python3
"""What this MI355X sustains for plain streaming, measured with library kernels on a 3.7 GB tensor — the practical ceiling
the roofline fractions of DESIGN.md are read against (8 TB/s is the spec, not reachable by any access pattern):
read-only (reduction), write-only (fill), copy (1 read : 1 write), and libatx's own streaming kernels beside them."""

from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    graft.load_package()
    from anemoi_transform_amd import native
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n = 6599680 * 140
    x = torch.rand(n, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    gb = n * 4 / 1e9
    rows = [
        ("torch sum (read only)", lambda: x.sum(), gb),
        ("torch fill_ (write only)", lambda: y.fill_(1.0), gb),
        ("torch copy_ (1 read : 1 write)", lambda: y.copy_(x), 2 * gb),
        ("torch add out= (1 read : 1 write)", lambda: torch.add(x, 1.0, out=y), 2 * gb),
        ("libatx stream_copy (1 read : 1 write)", lambda: native.stream_copy(x, y), 2 * gb),
        ("libatx reduce max (read only)", lambda: native.load().atx_reduce(x.data_ptr(), n, native.RED_MAX, red.data_ptr(), 0, None, 0, torch.cuda.current_stream().cuda_stream), gb),
    ]
    red = torch.zeros(1, dtype=torch.float64, device=dev)
    prog = native.level_program([[(native.OP_AFFINE, 0, 2.0, 1.0)] * 137], dev)
    xs, ys = x.view(6599680, 140), y.view(6599680, 140)
    rows.append(("libatx per-point affine (1 read : 1 write)", lambda: native.pointwise_stack(xs, ys, n_pts=6599680, n_lev=137, x_pitch=140, y_pitch=140,
                                                                                           layout=COLUMNS, prog=prog, n_stage=1), 2 * gb))
    for name, fn, bytes_gb in rows:
        ms, mn = bench.time_launches(fn, 20, 3)
        print(f"{name:48s} {ms:7.3f} ms  {bytes_gb / ms:6.2f} TB/s  ({bytes_gb / ms / 8:.3f} of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
