#!/usr/bin/env python3
"""Host-API cost on SMALL inputs (BASELINE config 2 shape: O96 -> 1 degree, a handful of host fields): where the time of
`regrid.forward(fields)` + `to_numpy()` goes when the kernels take microseconds."""

from __future__ import annotations

import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    graft.load_package()
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.fields import ArrayField, FieldList
    from anemoi_transform_amd.filters import create_filter_by_name
    from anemoi_transform_amd.grids import lookup

    torch.cuda.set_device(0)
    src, tgt = lookup("o96"), lookup([1.0, 1.0])
    matrix = interp.bilinear_octahedral(96, tgt)
    matrix = {**matrix, "out_latitudes": tgt["latitudes"], "out_longitudes": tgt["longitudes"]}
    rng = np.random.default_rng(0)
    regrid = create_filter_by_name("regrid", matrix=matrix)
    rescale = create_filter_by_name("rescale", scale=1.0, offset=-273.15, param="2t")
    for n_fields in (1, 10):
        fields = FieldList([ArrayField(280 + rng.standard_normal(len(src["latitudes"])), {"param": "2t", "levelist": l}, src["latitudes"],
                                       src["longitudes"]) for l in range(n_fields)])

        def run():
            out = (regrid | rescale).forward(fields)
            return [f.to_numpy(flatten=True) for f in out]

        for _ in range(3):
            run()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            run()
        dt = (time.perf_counter() - t0) / n
        print(f"{n_fields:3d} host field(s) O96 -> 1 deg, regrid | rescale, back to host: {dt * 1e3:.3f} ms per call", flush=True)
    # the small synchronous calls (one launch or three, one 16-byte read back): wall time per call on one 6.6 M-point field
    from anemoi_transform_amd import native

    dev = torch.device("cuda", 0)
    n = 6_599_680
    for tdt in (torch.float32, torch.float64):
        x = torch.randn(n, dtype=tdt, device=dev)
        for name, fn in (("reduce min", lambda: native.reduce(x, native.RED_MIN)), ("reduce min+max", lambda: native.reduce(x, native.RED_MINMAX)),
                         ("reduce nan-count", lambda: native.reduce(x, native.RED_NANCOUNT))):
            for _ in range(20):
                fn()
            t0 = time.perf_counter()
            for _ in range(500):
                fn()
            print(f"{name:18s} {str(tdt)[6:]:8s} one {n}-point field, result on the host: {(time.perf_counter() - t0) / 500 * 1e6:7.1f} us per call", flush=True)
    m = (torch.rand(n, device=dev) < 0.7).to(torch.uint8)
    for _ in range(20):
        native.mask_to_index(m, n)
    t0 = time.perf_counter()
    for _ in range(500):
        native.mask_to_index(m, n)
    print(f"mask_to_index      uint8    {n} points, 70 % kept, count on the host:        {(time.perf_counter() - t0) / 500 * 1e6:7.1f} us per call", flush=True)
    small = (torch.rand(65_160, device=dev) < 0.7).to(torch.uint8)
    for _ in range(20):
        native.mask_to_index(small)
    t0 = time.perf_counter()
    for _ in range(500):
        native.mask_to_index(small)
    print(f"mask_to_index      uint8    65160 points (1 degree grid):                      {(time.perf_counter() - t0) / 500 * 1e6:7.1f} us per call", flush=True)

    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        run()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
