#!/usr/bin/env python3
"""Does a GPU that sat idle run the first milliseconds of work slower?  The 8-rank strong-scaling line times 200 steps of 0.11 ms (22 ms) after
20 warm-up steps (2 ms) that follow seconds of host-only work.  Rank 0's shard of the headline: idle for `s` seconds, then W warm-up and 200
timed launches, for s in 0, 0.5, 2, 5 and W in 20, 200."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft, bench
graft.load_package()
from anemoi_transform_amd import native, interp
from anemoi_transform_amd.gather import GatherPlan, TARGET_COST_SHORT_LAUNCH
from anemoi_transform_amd.grids import lookup
from anemoi_transform_amd.stack import COLUMNS, Stack

dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
src, tgt = lookup('o1280'), lookup('0.25')
n_src, n_tgt, L = len(src['latitudes']), len(tgt['latitudes']), 137
idx, w = interp.knn_inverse_distance(src, tgt, k=4)
b = GatherPlan(n_src, n_tgt, index=idx, weights=w).bounds(8, target_cost=TARGET_COST_SHORT_LAUNCH)
lo, hi = b[0], b[1]
x = bench.synth_stack(src, L, torch.float64, dev, 0, COLUMNS)
out = Stack.empty(hi - lo, L, torch.float64, dev, COLUMNS)
i_d, w_d = torch.from_numpy(idx[lo:hi].astype(np.int32)).to(dev), torch.from_numpy(w[lo:hi]).to(dev)
step = lambda: native.regrid_ell(x.data, out.data, i_d, w_d, n_src=n_src, n_tgt=hi - lo, k=4, n_lev=L, src_pitch=x.pitch, out_pitch=out.pitch, layout=COLUMNS)
for idle in (0.0, 0.5, 2.0, 5.0, 0.0):
    for W in (20, 200):
        torch.cuda.synchronize(); time.sleep(idle)
        for _ in range(W): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): step()
        torch.cuda.synchronize()
        print(f"idle {idle:3.1f} s, {W:3d} warm-up steps: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per step", flush=True)
