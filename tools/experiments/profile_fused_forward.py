#!/usr/bin/env python3
"""Where the host time of one fused `regrid | orog_to_z | convert` call goes (cProfile, 137 device-resident O1280 fields)."""
from __future__ import annotations

import cProfile
import os
import pstats
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    graft.load_package()
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.fields import ArrayField, FieldList, new_field_from_stack
    from anemoi_transform_amd.filters import create_filter_by_name
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = 137
    src, tgt = lookup("o1280"), lookup("0.25")
    idx, w = interp.knn_inverse_distance(src, tgt, k=4, device=True, ties="index")
    matrix = {**interp.ell_to_csr(idx, w, len(src["latitudes"])), "out_latitudes": tgt["latitudes"], "out_longitudes": tgt["longitudes"]}
    x = bench.synth_stack(src, L, torch.float32, dev, 0, COLUMNS)
    template = ArrayField(np.zeros(1), {"param": "t"}, np.zeros(1), np.zeros(1))
    fields = FieldList([new_field_from_stack(x, l, template=template, latitudes=src["latitudes"], longitudes=src["longitudes"],
                                             metadata={"param": "t" if l < L - 1 else "orog", "levelist": l + 1}) for l in range(L)])
    chain = create_filter_by_name("regrid", matrix=matrix) | create_filter_by_name("orog_to_z") | create_filter_by_name("convert", unit_in="K", unit_out="degC", param="t")
    for _ in range(3):
        chain.forward(fields)
    torch.cuda.synchronize()
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(20):
        chain.forward(fields)
    torch.cuda.synchronize()
    prof.disable()
    stats = pstats.Stats(prof)
    stats.sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
