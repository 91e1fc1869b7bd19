#!/usr/bin/env python3
"""Thin stacks (1-13 levels): what the regrid launch costs in each storage form, next to a SIZE-MATCHED streaming bound.

For every (levels, k, dtype): the launch on a padded column stack (pitch rounded to 16 B), on a tight column stack
(pitch = levels: scalar kernels), on a field-major stack — and `atx_stream_copy` moving the same number of ALGORITHMIC
bytes (half read, half written), the practical ceiling for a transfer of that size on this device (launch ramp included).
O1280 -> 0.25 degree."""

from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    graft.load_package()
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS, FIELDS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    src, tgt = lookup("o1280"), lookup("0.25")
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    idx64, w64 = interp.knn_inverse_distance(src, tgt, k=4, device=True, ties="index")
    U = {4: int(np.unique(idx64).size), 1: int(np.unique(idx64[:, 0]).size)}
    tiny = torch.zeros(4096, dtype=torch.uint8, device=dev)
    tiny2 = torch.zeros_like(tiny)
    ms, mn = bench.time_launches(lambda: native.stream_copy(tiny, tiny2), 50, 5)
    print(f"launch floor (4 KB stream copy between HIP events): avg {ms * 1e3:.1f} us, min {mn * 1e3:.1f} us", flush=True)
    for tdt, npdt, B, tag in ((torch.float32, np.float32, 4, "f32"), (torch.float64, np.float64, 8, "f64")):
        per16 = 16 // B
        for L in (1, 2, 3, 4, 8, 13):
            data = torch.randn(n_src, L, dtype=tdt, device=dev)
            for k in (4, 1):
                idx = torch.from_numpy(np.ascontiguousarray(idx64[:, :k]).astype(np.int32)).to(dev)
                w = torch.from_numpy(w64.astype(npdt)).to(dev) if k > 1 else None
                alg = bench.algorithmic_bytes(L, B, U[k], n_tgt, k)
                half = (alg // 2 + 15) // 16 * 16
                a, b = torch.empty(half, dtype=torch.uint8, device=dev), torch.empty(half, dtype=torch.uint8, device=dev)
                t_copy, _ = bench.time_launches(lambda: native.stream_copy(a, b), 30, 5)
                del a, b
                line = f"{tag} L={L:2d} k={k}: alg {alg / 1e6:7.1f} MB; size-matched copy {t_copy * 1e3:6.1f} us"
                forms = {}
                pitch_pad = (L + per16 - 1) // per16 * per16
                for name, layout, pitch in (("padded", COLUMNS, pitch_pad), ("tight", COLUMNS, L), ("fields", FIELDS, None)):
                    if name == "tight" and pitch == pitch_pad:
                        continue
                    if layout == COLUMNS:
                        x = torch.zeros(n_src, pitch, dtype=tdt, device=dev)
                        x[:, :L] = data
                        out = torch.empty(n_tgt, pitch, dtype=tdt, device=dev)
                        sp, op = pitch, pitch
                    else:
                        x = data.T.contiguous()
                        out = torch.empty(L, n_tgt, dtype=tdt, device=dev)
                        sp, op = n_src, n_tgt
                    t, _ = bench.time_launches(lambda: native.regrid_ell(x, out, idx, w, n_src=n_src, n_tgt=n_tgt, k=k, n_lev=L,
                                                                         src_pitch=sp, out_pitch=op, layout=layout), 30, 5)
                    forms[name] = t
                    res = out[:, :L] if layout == COLUMNS else out.T
                    if "ref" not in forms:
                        forms["ref"] = res.clone()
                    else:
                        assert torch.equal(res, forms["ref"]), (name, L, k)
                    line += f" | {name} {t * 1e3:6.1f} us ({alg / (t * 1e-3) / 8e12:.3f} of 8 TB/s, {t_copy / t:.2f} of copy)"
                    del x, out
                print(line, flush=True)
            del data
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
