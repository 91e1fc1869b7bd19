#!/usr/bin/env python3
"""Fixed k between the direct kernel's compile-time widths (9-11, 13-15): GatherPlan pads them to 12 / 16 with absent entries — time and
bits against the tiled run-time-k kernel on the table as given (O1280 -> 0.25 deg, 137 levels, targets in the policy's order)."""
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
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.gather import GatherPlan, target_order_for
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = 137
    src, tgt = lookup("o1280"), lookup("0.25")
    n, nt = len(src["latitudes"]), len(tgt["latitudes"])
    idx16, w16 = interp.knn_inverse_distance(src, tgt, k=16, device=True, ties="index")
    for tdt, B, npd, tag in ((torch.float32, 4, np.float32, "f32"), (torch.float64, 8, np.float64, "f64")):
        x = bench.synth_stack(src, L, tdt, dev, 0, COLUMNS)
        for k in (10, 13):
            idx, w = idx16[:, :k], w16[:, :k] / w16[:, :k].sum(axis=1, keepdims=True)
            plan = GatherPlan(n, nt, index=idx, weights=w)
            assert plan.padded and plan.k in (12, 16)
            plan.order_targets(target_order_for(tgt["latitudes"], tgt["longitudes"], k))
            out = plan.apply(x)
            ref = x.new_like(n_pts=nt)
            idx_d = torch.from_numpy(np.ascontiguousarray(idx).astype(np.int32).reshape(-1)).to(dev)
            w_d = torch.from_numpy(np.ascontiguousarray(w).astype(npd).reshape(-1)).to(dev)
            kw = dict(n_src=n, n_tgt=nt, k=k, n_lev=L, src_pitch=x.pitch, out_pitch=ref.pitch, layout=COLUMNS)
            native.regrid_ell(x.data, ref.data, idx_d, w_d, **kw)
            assert torch.equal(out.data[:, :L], ref.data[:, :L])
            alg = bench.algorithmic_bytes(L, B, int(np.unique(idx).size), nt, k)
            ms_p = launches(lambda: plan.apply(x))
            ms_t = launches(lambda: native.regrid_ell(x.data, ref.data, idx_d, w_d, **kw))
            print(f"{tag} k={k}: padded to {plan.k} on the direct kernel, ordered {ms_p:.3f} ms ({alg / (ms_p * 1e-3) / 8e12:.3f}) | "
                  f"as given on the tiled kernel, natural order {ms_t:.3f} ms ({alg / (ms_t * 1e-3) / 8e12:.3f}); same bits", flush=True)
        del x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
