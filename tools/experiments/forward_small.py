#!/usr/bin/env python3
"""Wall time of Filter.forward() on SMALL device-resident FieldLists (1, 4, 13 fields of O96 / O1280): where the host side, not the kernel, is the cost."""
from __future__ import annotations

import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from anemoi_transform_amd.fields import fieldlist_from_dicts  # noqa: E402
from anemoi_transform_amd.filters import create_filter_by_name  # noqa: E402
from anemoi_transform_amd.grids import lookup  # noqa: E402
from anemoi_transform_amd.prefetch import to_device  # noqa: E402


def main():
    torch.cuda.set_device(0)
    for grid, out in (("o96", [1.0, 1.0]), ("o1280", "0.25")):
        g = lookup(grid)
        n = len(g["latitudes"])
        rng = np.random.default_rng(0)
        regrid = create_filter_by_name("regrid", in_grid=grid, out_grid=out, method="nearest")
        rescale = create_filter_by_name("rescale", scale=2.0, offset=1.0, param="t")
        for n_fields in (1, 4, 13):
            specs = [{"param": "t", "levelist": l, "values": rng.standard_normal(n).astype(np.float32), "latitudes": g["latitudes"], "longitudes": g["longitudes"]}
                     for l in range(n_fields)]
            dev_fl = to_device(fieldlist_from_dicts(specs))
            for name, f in (("regrid", regrid), ("rescale", rescale), ("regrid|rescale", regrid | rescale)):
                for _ in range(5):
                    f.forward(dev_fl)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                reps = 200
                for _ in range(reps):
                    f.forward(dev_fl)
                torch.cuda.synchronize()
                print(f"{grid:6s} {n_fields:2d} field(s) {name:16s} {(time.perf_counter() - t0) / reps * 1e6:8.1f} us per forward()")
    if "--profile" in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(300):
            regrid.forward(dev_fl)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(25)


if __name__ == "__main__":
    main()
