#!/usr/bin/env python3
"""Regrid k=4 / k=1 launch time against the number of levels in the stack (surface fields, pressure-level stacks of 13 / 37,
60 / 137 model levels), with the BOUND each case is measured against.  Column stacks, O1280 -> 0.25 degree.

For thick stacks the bound is the HBM roofline on ALGORITHMIC bytes (8 TB/s peak).  Thin stacks cannot be priced that way:
  * a launch between two HIP events costs ~11 us before it moves a byte (a 4 KB copy), and
  * a column of 1-13 float32 levels is shorter than the 64-byte sector HBM is read in, so the compulsory traffic is the set
    of distinct SECTORS the index table touches (counted here on the host), not the referenced elements.
The stated bound for a thin case is therefore MEASURED: `atx_stream_copy` moving the same number of compulsory bytes (half
read, half written) in one launch — the time this device needs to move that many bytes at all.  Reported per case: launch
time; fraction of 8 TB/s on algorithmic bytes; compulsory / algorithmic bytes; size-matched copy time and the fraction
of it the regrid launch reaches (1.0 = as fast as a plain copy of the compulsory bytes).  `--tiles` adds the tile sweep of
the tiled kernel (round 1's check of its heuristic)."""

from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402

SECTOR = 64


def compulsory_bytes(idx: np.ndarray, n_src: int, n_tgt: int, L: int, B: int, pitch: int, k: int) -> int:
    """Distinct 64-byte sectors of the source stack the table touches + the output rows as stored + one pass over the tables."""
    rows = np.unique(idx)
    start = rows.astype(np.int64) * (pitch * B)
    first, last = start // SECTOR, (start + L * B - 1) // SECTOR
    touched = np.zeros(n_src * pitch * B // SECTOR + 2, dtype=bool)
    for s in range(int((last - first).max()) + 1):
        sel = first + s <= last
        touched[(first + s)[sel]] = True
    out_bytes = n_tgt * pitch * B
    tables = n_tgt * k * 4 + (n_tgt * k * B if k > 1 else 0)
    return int(touched.sum()) * SECTOR + out_bytes + tables


def main():
    graft.load_package()
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS, Stack

    tiles = "--tiles" in sys.argv
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    src, tgt = lookup("o1280"), lookup("0.25")
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    idx64, w64 = interp.knn_inverse_distance(src, tgt, k=4, device=True, ties="index")
    U = {4: int(np.unique(idx64).size), 1: int(np.unique(idx64[:, 0]).size)}
    tiny = torch.zeros(4096, dtype=torch.uint8, device=dev)
    floor_ms, _ = bench.time_launches(lambda: native.stream_copy(tiny, torch.empty_like(tiny)), 50, 5)
    print(f"launch floor between HIP events (4 KB atx_stream_copy): {floor_ms * 1e3:.1f} us", flush=True)
    for tdt, npdt, B, tag in ((torch.float32, np.float32, 4, "f32"), (torch.float64, np.float64, 8, "f64")):
        for L in (1, 4, 13, 37, 60, 137):
            x = bench.synth_stack(src, L, tdt, dev, 0, COLUMNS)
            out = Stack.empty(n_tgt, L, tdt, dev, COLUMNS)
            for k in (4, 1):
                table = np.ascontiguousarray(idx64[:, :k])
                idx = torch.from_numpy(table.astype(np.int32)).to(dev)
                w = torch.from_numpy(w64.astype(npdt)).to(dev) if k > 1 else None
                alg = bench.algorithmic_bytes(L, B, U[k], n_tgt, k)
                comp = compulsory_bytes(table, n_src, n_tgt, L, B, x.pitch, k)
                half = (comp // 2 + 15) // 16 * 16
                a, b = torch.empty(half, dtype=torch.uint8, device=dev), torch.empty(half, dtype=torch.uint8, device=dev)
                t_copy, _ = bench.time_launches(lambda: native.stream_copy(a, b), 30, 5)
                del a, b

                def launch():
                    native.regrid_ell(x.data, out.data, idx, w, n_src=n_src, n_tgt=n_tgt, k=k, n_lev=L, src_pitch=x.pitch, out_pitch=out.pitch,
                                      layout=COLUMNS)

                auto, _ = bench.time_launches(launch, 30, 5)
                line = (f"{tag} L={L:3d} k={k} pitch={x.pitch:3d}: {auto * 1e3:7.1f} us | {alg / (auto * 1e-3) / 8e12:.3f} of 8 TB/s on algorithmic bytes "
                        f"({alg / 1e6:7.1f} MB) | compulsory sectors {comp / alg:4.2f}x algorithmic -> {comp / (auto * 1e-3) / 1e12:4.2f} TB/s | "
                        f"size-matched copy {t_copy * 1e3:6.1f} us -> {t_copy / auto:.2f} of the stated bound")
                if tiles:
                    best = None
                    for tile in (8, 16, 32, 64, 128, 256):
                        native.set_tuning(tile)
                        ms, _ = bench.time_launches(launch, 20, 3)
                        line += f" | tile {tile}: {ms * 1e3:.1f}"
                        if best is None or ms < best[1]:
                            best = (tile, ms)
                    native.set_tuning(0)
                    line += f" | auto/best tiled = {auto / best[1]:.2f}"
                print(line, flush=True)
            del x, out
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
