#!/usr/bin/env python3
"""Where the time of the host-fed path goes: 137 float32 O1280 fields in host memory -> regrid filter -> 137 host arrays."""
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


def main():
    graft.load_package()
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.fields import ArrayField, FieldList
    from anemoi_transform_amd.filters import create_filter_by_name
    from anemoi_transform_amd.grids import lookup

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = 137
    src, tgt = lookup("o1280"), lookup("0.25")
    n_src = len(src["latitudes"])
    idx, w = interp.knn_inverse_distance(src, tgt, k=4, device=True, ties="index")
    matrix = {**interp.ell_to_csr(idx, w, n_src), "out_latitudes": tgt["latitudes"], "out_longitudes": tgt["longitudes"]}
    rng = np.random.default_rng(3)
    host = [(280 + rng.standard_normal(n_src)).astype(np.float32) for _ in range(L)]
    fields = FieldList([ArrayField(a, {"param": "t", "levelist": l + 1}, src["latitudes"], src["longitudes"]) for l, a in enumerate(host)])
    regrid = create_filter_by_name("regrid", matrix=matrix)
    for _ in range(2):
        result = regrid.forward(fields)
        arrays = [f.to_numpy(flatten=True) for f in result]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    result = regrid.forward(fields)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    arrays = [f.to_numpy(flatten=True) for f in result]
    t2 = time.perf_counter()
    print(f"forward (upload + launch) {1e3 * (t1 - t0):.1f} ms, to_numpy of {L} fields {1e3 * (t2 - t1):.1f} ms")
    prof = cProfile.Profile()
    prof.enable()
    result = regrid.forward(fields)
    arrays = [f.to_numpy(flatten=True) for f in result]
    prof.disable()
    pstats.Stats(prof).sort_stats("cumulative").print_stats(25)
    del arrays


if __name__ == "__main__":
    main()
