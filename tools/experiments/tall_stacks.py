#!/usr/bin/env python3
"""Does the gather get more out of HBM when SEVERAL variables share a column?  O1280 -> 0.25 degree, k = 4: one launch over a stack of
137 / 274 / 411 / 548 / 822 levels (1 .. 6 variables x 137 levels in one column of 0.5 .. 3.3 KB) against the batched launch over that many
separate 137-level stacks (what bench.py's config 4 does).  A longer column wastes less of its first and last 128-byte line."""
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
    from anemoi_transform_amd.stack import COLUMNS, Stack

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    src, tgt = lookup("o1280"), lookup(sys.argv[1] if len(sys.argv) > 1 else "0.25")
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    idx64, w64 = interp.knn_inverse_distance(src, tgt, k=4, device=True, ties="index")
    U = int(np.unique(idx64).size)
    idx = torch.from_numpy(idx64.astype(np.int32)).to(dev)
    for tdt, npdt, B, tag in ((torch.float32, np.float32, 4, "f32"), (torch.float64, np.float64, 8, "f64")):
        w = torch.from_numpy(w64.astype(npdt)).to(dev)
        for n_var in (1, 2, 3, 4, 6):
            L = 137 * n_var
            x = Stack.empty(n_src, L, tdt, dev, COLUMNS, zero=True)
            x.data[:, :L].normal_(250.0, 20.0)
            out = Stack.empty(n_tgt, L, tdt, dev, COLUMNS)
            tall, _ = bench.time_launches(lambda: native.regrid_ell(x.data, out.data, idx, w, n_src=n_src, n_tgt=n_tgt, k=4, n_lev=L, src_pitch=x.pitch,
                                                                    out_pitch=out.pitch, layout=COLUMNS), 10, 2)
            alg = bench.algorithmic_bytes(L, B, U, n_tgt, 4)
            del x, out
            xs = [Stack.empty(n_src, 137, tdt, dev, COLUMNS, zero=True) for _ in range(n_var)]
            for s in xs:
                s.data[:, :137].normal_(250.0, 20.0)
            outs = [Stack.empty(n_tgt, 137, tdt, dev, COLUMNS) for _ in range(n_var)]
            batched, _ = bench.time_launches(lambda: native.regrid_ell_batch([s.data for s in xs], [o.data for o in outs], idx, w, n_src=n_src, n_tgt=n_tgt, k=4,
                                                                              n_lev=137, src_pitch=xs[0].pitch, out_pitch=outs[0].pitch, layout=COLUMNS), 10, 2)
            alg_b = n_var * bench.algorithmic_bytes(137, B, U, n_tgt, 4)
            print(f"{tag} {n_var} variable(s) x 137 levels: ONE stack of {L} levels {tall:7.3f} ms = {alg / tall / 1e6 / 8000:.3f} | "
                  f"{n_var} stacks, batched launch {batched:7.3f} ms = {alg_b / batched / 1e6 / 8000:.3f}", flush=True)
            del xs, outs
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
