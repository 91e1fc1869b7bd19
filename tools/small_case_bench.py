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
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        run()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
