#!/usr/bin/env python3
"""The host-fed path end to end: 137 float32 O1280 fields in host memory -> regrid filter -> 137 host arrays.
Reports upload (H2D + relayout), kernel, download (relayout + D2H) and the whole `forward` + `to_numpy` wall time —
the PCIe-inclusive rate DESIGN.md quotes next to the resident-data headline."""

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


def main():
    graft.load_package()
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.fields import ArrayField, FieldList
    from anemoi_transform_amd.filters import create_filter_by_name
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import Stack

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 137
    src, tgt = lookup("o1280"), lookup("0.25")
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    idx, w = interp.knn_inverse_distance(src, tgt, k=4, device=True, ties="index")
    matrix = {**interp.ell_to_csr(idx, w, n_src), "out_latitudes": tgt["latitudes"], "out_longitudes": tgt["longitudes"]}
    rng = np.random.default_rng(3)
    host = [(280 + rng.standard_normal(n_src)).astype(np.float32) for _ in range(L)]
    fields = FieldList([ArrayField(a, {"param": "t", "levelist": l + 1}, src["latitudes"], src["longitudes"]) for l, a in enumerate(host)])
    regrid = create_filter_by_name("regrid", matrix=matrix)

    def sync():
        torch.cuda.synchronize()

    res = {}
    for rep in range(3):
        sync(); t0 = time.perf_counter()
        st = Stack.from_fields(host, dev=dev)
        sync(); t1 = time.perf_counter()
        out = regrid.interpolator.plan.apply(st)
        sync(); t2 = time.perf_counter()
        back = out.numpy()
        t3 = time.perf_counter()
        res = {"upload_ms": (t1 - t0) * 1e3, "kernel_ms": (t2 - t1) * 1e3, "download_ms": (t3 - t2) * 1e3}
    in_gb, out_gb = L * n_src * 4 / 1e9, L * n_tgt * 4 / 1e9
    res["upload_GBs"] = in_gb / (res["upload_ms"] * 1e-3)
    res["download_GBs"] = out_gb / (res["download_ms"] * 1e-3)
    for rep in range(2):
        sync(); t0 = time.perf_counter()
        result = regrid.forward(fields)
        arrays = [f.to_numpy(flatten=True) for f in result]
        t1 = time.perf_counter()
    res["filter_forward_plus_to_numpy_ms"] = (t1 - t0) * 1e3
    res["host_fed_grid_points_per_s"] = L * n_tgt / (t1 - t0)
    assert np.array_equal(arrays[-1], back[-1])

    # a job of several FieldLists (dates): one after the other, and with the next upload running ahead (prefetch_to_device)
    from anemoi_transform_amd.prefetch import prefetch_to_device

    n_lists = 6
    jobs = [fields] + [FieldList([ArrayField(a + np.float32(d), {"param": "t", "levelist": l + 1}, src["latitudes"], src["longitudes"])
                                  for l, a in enumerate(host)]) for d in range(1, n_lists)]

    def consume(result):
        return [f.to_numpy(flatten=True) for f in result]

    sync(); t0 = time.perf_counter()
    last = None
    for fl in jobs:
        last = consume(regrid.forward(fl))
    sync(); t_seq = time.perf_counter() - t0
    sync(); t0 = time.perf_counter()
    last_p = None
    for dev_fl in prefetch_to_device(iter(jobs), depth=1):
        last_p = consume(regrid.forward(dev_fl))
    sync(); t_pre = time.perf_counter() - t0
    assert np.array_equal(last[-1], last_p[-1])
    res["job_of_6_lists_sequential_ms_per_list"] = t_seq / n_lists * 1e3
    res["job_of_6_lists_prefetch_ms_per_list"] = t_pre / n_lists * 1e3
    res["job_host_fed_grid_points_per_s_prefetch"] = n_lists * L * n_tgt / t_pre
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
