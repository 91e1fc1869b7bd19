#!/usr/bin/env python3
"""Wall time of Filter.forward() for every GPU-backed filter on HBM-resident FieldLists of O1280 fields (float32), next to the time
the bytes it must move would take at 6 TB/s — host overhead (grouping, field wrappers, operand assembly) shows as the gap."""
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


def wall(fn, n=5, warm=2):
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
    from anemoi_transform_amd.fields import ArrayField, FieldList, new_field_from_stack
    from anemoi_transform_amd.filters import create_filter_by_name
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = 137
    src = lookup("o1280")
    n = len(src["latitudes"])
    template = ArrayField(np.zeros(1), {"param": "t", "levelist": 1, "levtype": "ml"}, np.zeros(1), np.zeros(1), mars=True)
    stacks = {}

    def fieldlist(params, scale=1.0, shift=0.0, levels=L, seed=0, nan_frac=0.0):
        """`levels` fields of each param in `params`, one stack per param."""
        out = []
        for i, p in enumerate(params):
            st = bench.synth_stack(src, levels, torch.float32, dev, seed + i, COLUMNS)
            if scale != 1.0 or shift != 0.0:
                st.data.mul_(scale).add_(shift)
            if nan_frac:
                st.data[::int(1 / nan_frac)] = float("nan")
            stacks[(p, seed)] = st
            out += [new_field_from_stack(st, l, template=template, latitudes=src["latitudes"], longitudes=src["longitudes"],
                                         metadata={"param": p, "levelist": l + 1, "levtype": "ml"}) for l in range(levels)]
        return FieldList(out)

    stack_bytes = n * L * 4
    t = fieldlist(["t"])
    res = {}

    def record(name, fn, bytes_moved):
        ms = wall(fn)
        ideal = bytes_moved / 6e12 * 1e3
        res[name] = {"ms": ms, "ideal_ms_at_6TBs": ideal}
        print(f"{name:58s} {ms:8.3f} ms wall   ({ideal:6.3f} ms of traffic at 6 TB/s)", flush=True)

    record("rescale (137 fields of t)", lambda: create_filter_by_name("rescale", scale=1.0, offset=-273.15, param="t").forward(t), 2 * stack_bytes)
    f = create_filter_by_name("rescale", scale=1.0, offset=-273.15, param="t")
    record("rescale, filter built once", lambda: f.forward(t), 2 * stack_bytes)
    f = create_filter_by_name("convert", unit_in="K", unit_out="degC", param="t")
    record("convert K -> degC", lambda: f.forward(t), 2 * stack_bytes)
    f = create_filter_by_name("clip_fields", minimum=250.0, maximum=300.0, param="t") if False else None
    try:
        f = create_filter_by_name("clip_fields", param="t", minimum=250.0, maximum=300.0)
        record("clip", lambda: f.forward(t), 2 * stack_bytes)
    except Exception as e:  # constructor spelling differs: report, keep going
        print("clip: ", repr(e))
    tn = fieldlist(["t"], seed=40, nan_frac=0.01)
    try:
        f = create_filter_by_name("impute_nans_fields", param="t", value=0.0)
        record("impute_nans", lambda: f.forward(tn), 2 * stack_bytes)
    except Exception as e:
        print("impute_nans: ", repr(e))
    lnsp = fieldlist(["lnsp"], scale=0.04, levels=1, seed=3)
    f = create_filter_by_name("lnsp_to_sp")
    record("lnsp_to_sp (1 field)", lambda: f.forward(lnsp), 2 * n * 4)
    oro = fieldlist(["orog"], levels=1, seed=5)
    f = create_filter_by_name("orog_to_z_fields")
    record("orog_to_z (1 field)", lambda: f.forward(oro), 2 * n * 4)
    try:
        mask_field = fieldlist(["lsm"], scale=1.0 / 320.0, levels=1, seed=7)
        both = FieldList(list(t) + list(mask_field))
        f = create_filter_by_name("apply_mask_fields", mask_param="lsm", threshold=0.85, threshold_operator=">")
        record("apply_mask from a field of the stream (137 + 1 fields)", lambda: f.forward(both), 2 * stack_bytes)
    except Exception as e:
        print("apply_mask: ", repr(e))
    try:
        f = create_filter_by_name("remove_nans_fields")
        record("remove_nans (137 fields, 1 % NaN)", lambda: f.forward(tn), 2 * stack_bytes)
    except Exception as e:
        print("remove_nans: ", repr(e))
    sn = fieldlist(["sd", "rsn"], levels=L, seed=10)
    f = create_filter_by_name("snow_cover")
    record("snow_cover (137 pairs)", lambda: f.forward(sn), 3 * stack_bytes)
    f = create_filter_by_name("snow_depth_m")
    record("snow_depth_m (137 pairs)", lambda: f.forward(sn), 3 * stack_bytes)
    ang = fieldlist(["mwd"], scale=1.0, shift=-100.0, levels=L, seed=20)
    f = create_filter_by_name("cos_sin_mean_wave_direction")
    record("cos_sin_mean_wave_direction (137 fields)", lambda: f.forward(ang), 3 * stack_bytes)
    uv = fieldlist(["u", "v"], scale=0.1, shift=-27.0, levels=L, seed=30)
    f = create_filter_by_name("uv_to_ddff")
    record("uv_to_ddff (137 pairs)", lambda: f.forward(uv), 4 * stack_bytes)
    f = create_filter_by_name("sum", params=["u", "v"], output="uv")
    record("sum of two params (137 pairs)", lambda: f.forward(uv), 3 * stack_bytes)
    wtq = fieldlist(["w", "t", "q"], scale=0.01, levels=L, seed=50)
    f = create_filter_by_name("w_to_wz")
    record("w_to_wz (137 triples)", lambda: f.forward(wtq), 4 * stack_bytes)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
