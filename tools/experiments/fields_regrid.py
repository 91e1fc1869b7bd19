#!/usr/bin/env python3
"""Regrid entry points on FIELD-MAJOR stacks (O1280 -> 0.25 degrees, 137 fields): fixed k, ragged CSR, with a fused program."""
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
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import FIELDS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = 137
    src, tgt = lookup("o1280"), lookup("0.25")
    n, nt = len(src["latitudes"]), len(tgt["latitudes"])
    idx, w = interp.knn_inverse_distance(src, tgt, k=4)
    keep = (np.arange(idx.size) % 9 != 0).reshape(idx.shape)
    ptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))])
    print("library:", native.lib_path(), flush=True)
    for tdt, B, tag in ((torch.float32, 4, "f32"), (torch.float64, 8, "f64")):
        x = bench.synth_stack(src, L, tdt, dev, 0, FIELDS)
        plans = {
            "k=4": (GatherPlan(n, nt, index=idx, weights=w), bench.algorithmic_bytes(L, B, int(np.unique(idx).size), nt, 4)),
            "k=1": (GatherPlan(n, nt, index=idx[:, 0]), bench.algorithmic_bytes(L, B, int(np.unique(idx[:, 0]).size), nt, 1)),
            "csr ragged(3-4)": (GatherPlan(n, nt, csr=(w[keep], idx[keep], ptr)),
                                L * B * (int(np.unique(idx[keep]).size) + nt) + int(keep.sum()) * (4 + B) + 4 * nt),
            "ragged(3-4) padded": (GatherPlan.from_matrix(dict(matrix_data=w[keep], matrix_indices=idx[keep], matrix_indptr=ptr, matrix_shape=(nt, n))),
                                   L * B * (int(np.unique(idx[keep]).size) + nt) + int(keep.sum()) * (4 + B) + 4 * nt),
        }
        idx16, w16 = interp.knn_inverse_distance(src, tgt, k=16, device=True, ties="index")
        keep16 = np.random.default_rng(16).random(idx16.shape) < 0.75
        keep16[:, :9] = True
        ptr16 = np.concatenate([[0], np.cumsum(keep16.sum(axis=1))])
        plans["csr rows of 9-16"] = (GatherPlan(n, nt, csr=(w16[keep16], idx16[keep16], ptr16)),
                                     L * B * (int(np.unique(idx16[keep16]).size) + nt) + int(keep16.sum()) * (4 + B) + 4 * nt)
        plans["k=16"] = (GatherPlan(n, nt, index=idx16, weights=w16), bench.algorithmic_bytes(L, B, int(np.unique(idx16).size), nt, 16))
        flagged = GatherPlan(n, nt, index=idx, weights=w)
        flagged.padded = True  # the same table through the padded-row instantiation: nothing is absent
        plans["k=4 as a padded table"] = (flagged, bench.algorithmic_bytes(L, B, int(np.unique(idx).size), nt, 4))
        prog = native.level_program([[(native.OP_AFFINE, 0, 1.0 + 0.001 * l, -273.15) for l in range(L)]], dev)
        for name, (plan, alg) in plans.items():
            ms = launches(lambda: plan.apply(x))
            ms_e = launches(lambda: plan.apply(x, prog=prog, n_stage=1))
            print(f"{tag} fields regrid {name:20s} {ms:7.3f} ms  {alg / (ms * 1e-3) / 8e12:.3f}   + a scale per field {ms_e:7.3f} ms", flush=True)
        del x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
