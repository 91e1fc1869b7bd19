#!/usr/bin/env python3
"""Fused regrid epilogue (K10, config 5): launch time of the three routes the operators can take to the gather kernel —
by value in the kernel arguments (uniform program), the host-built per-vector table in global memory, the tiled kernel's
per-workgroup LDS table — next to the plain gather, interleaved.  Column stacks, k = 4.

    python tools/epilogue_routes.py [o1280|o2560] [reps]
"""

from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    graft.load_package()
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS, Stack

    grid = sys.argv[1] if len(sys.argv) > 1 else "o1280"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    src, tgt = lookup(grid), lookup("0.25")
    n_src, n_tgt, L = len(src["latitudes"]), len(tgt["latitudes"]), 137
    idx64, w64 = interp.knn_inverse_distance(src, tgt, k=4, device=True, ties="index")
    U = int(np.unique(idx64).size)
    idx = torch.from_numpy(idx64.astype(np.int32)).to(dev)
    mask = torch.from_numpy((np.random.default_rng(0).random(n_tgt) < 0.3).astype(np.uint8)).to(dev)
    mul, aff, cp = (native.OP_MUL, 0, 9.80665, 0.0), (native.OP_AFFINE, 0, 1.0, -273.15), (native.OP_COPY, 0, 0.0, 0.0)
    programs = {
        "uniform 2 stages (orog_to_z, convert on all levels)": ([[mul] * L, [aff] * L], None),
        "config 5 (136 levels convert, 1 level orog_to_z)": ([[cp] * (L - 1) + [mul], [aff] * (L - 1) + [cp]], None),
        "config 5 + apply_mask stage": ([[cp] * (L - 1) + [mul], [aff] * (L - 1) + [cp], [(native.OP_COPY, 1, 0.0, 0.0)] * L], mask),
    }
    for tdt, npdt, B, tag in ((torch.float32, np.float32, 4, "f32"), (torch.float64, np.float64, 8, "f64")):
        x = Stack.empty(n_src, L, tdt, dev, COLUMNS, zero=True)
        x.data[:, :L].normal_(280.0, 20.0)
        out = Stack.empty(n_tgt, L, tdt, dev, COLUMNS)
        w = torch.from_numpy(w64.astype(npdt)).to(dev)
        alg = bench.algorithmic_bytes(L, B, U, n_tgt, 4)

        def launcher(prog, n_stage, m):
            return lambda: native.regrid_ell(x.data, out.data, idx, w, n_src=n_src, n_tgt=n_tgt, k=4, n_lev=L, src_pitch=x.pitch,
                                             out_pitch=out.pitch, layout=COLUMNS, prog=prog, n_stage=n_stage, tgt_mask=m)

        for name, (stages, m) in programs.items():
            routes = {"plain gather (no epilogue)": launcher(None, 0, None)}
            full = native.level_program(stages, dev)
            table = native.level_program(stages, dev)
            table.host_prog = None
            tiled = native.level_program(stages, dev)
            tiled.host_prog, tiled.vec_prog = None, {}
            routes["library choice (host_prog + vec_prog given)"] = launcher(full, len(stages), m)
            routes["per-vector table in global memory"] = launcher(table, len(stages), m)
            routes["tiled kernel, LDS table per workgroup"] = launcher(tiled, len(stages), m)
            times = {r: [] for r in routes}
            for _ in range(reps):  # interleaved: drift hits every route alike
                for r, fn in routes.items():
                    times[r].append(bench.time_launches(fn, 20, 3)[0])
            print(f"{grid} {tag} | {name}")
            for r, t in times.items():
                ms = float(np.median(t))
                print(f"    {r:48s} {ms * 1e3:7.1f} us   {alg / (ms * 1e-3) / 8e12:.3f} of 8 TB/s on algorithmic bytes", flush=True)
        del x, out, w
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
