#!/usr/bin/env python3
"""Filter-API overhead on HBM-resident fields: Filter.forward() on a FieldList of 137 fields (one O1280 stack)
vs the bare kernel launch — what the Python host layer (grouping, programs, field wrappers) costs per call."""

from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def wall(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


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
    regrid = create_filter_by_name("regrid", matrix=matrix)
    chain = regrid | create_filter_by_name("orog_to_z") | create_filter_by_name("convert", unit_in="K", unit_out="degC", param="t")
    rescale = create_filter_by_name("rescale", scale=1.0, offset=-273.15, param="t")
    # config-4 shape: several variables x 137 levels on the same grid pair -> one batched launch (atx_regrid_ell_batch)
    more = [bench.synth_stack(src, L, torch.float32, dev, s, COLUMNS) for s in (1, 2)]
    many = FieldList(list(fields) + [
        new_field_from_stack(st, l, template=template, latitudes=src["latitudes"], longitudes=src["longitudes"],
                             metadata={"param": name, "levelist": l + 1})
        for st, name in zip(more, ("u", "v")) for l in range(L)])
    res = {
        "regrid.forward (137 fields, one stack)": wall(lambda: regrid.forward(fields)),
        "regrid.forward (411 fields, three stacks, one batched launch)": wall(lambda: regrid.forward(many)),
        "fused pipeline regrid|orog_to_z|convert": wall(lambda: chain.forward(fields)),
        "rescale.forward on 136 of 137 source fields": wall(lambda: rescale.forward(fields)),
    }
    for k, v in res.items():
        print(f"{k:50s} {v:8.3f} ms wall per call")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
