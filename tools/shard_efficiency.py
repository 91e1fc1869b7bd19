#!/usr/bin/env python3
"""Per-shard cost of the target-sharded regrid on ONE GPU: time of `world` launches (one per stack) for
every rank's shard, with equal-count and with traffic-balanced shard boundaries.  The slowest rank sets the
multi-GPU step time, so max/mean over ranks bounds the weak-scaling efficiency."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft, bench
graft.load_package()
from anemoi_transform_amd import native, interp
from anemoi_transform_amd.gather import GatherPlan
from anemoi_transform_amd.grids import lookup
from anemoi_transform_amd.stack import COLUMNS, Stack

dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
src, tgt = lookup('o1280'), lookup('0.25')
n_src, n_tgt, L = len(src['latitudes']), len(tgt['latitudes']), 137
idx, w = interp.knn_inverse_distance(src, tgt, k=4, device=True, ties="index")
plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
TDT, NPDT = (torch.float64, np.float64) if os.environ.get("ATX_SHARD_DTYPE", "f64") == "f64" else (getattr(torch, "float32"), getattr(np, "float32"))  # f64: the headline since round 3
if len(sys.argv) > 2:  # try another weight of the shard cost model
    from anemoi_transform_amd import gather as _gather

    _gather.TARGET_COST = float(sys.argv[2])
    print('TARGET_COST =', _gather.TARGET_COST)
stacks = [bench.synth_stack(src, L, TDT, dev, s, COLUMNS) for s in range(world)]
for name, bounds, batched in (("equal-count", [(n_tgt * r) // world for r in range(world + 1)], False),
                              ("traffic-balanced", plan.bounds(world), False),
                              ("balanced+batched", plan.bounds(world), True)):
    times = []
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        idx_d = torch.from_numpy(idx[lo:hi].astype(np.int32)).to(dev); w_d = torch.from_numpy(w[lo:hi].astype(NPDT)).to(dev)
        outs = [Stack.empty(hi - lo, L, TDT, dev, COLUMNS) for _ in range(world)]
        def step():
            if batched:  # one launch over the `world` stacks of the step (atx_regrid_ell_batch)
                native.regrid_ell_batch([s.data for s in stacks], [o.data for o in outs], idx_d, w_d, n_src=n_src, n_tgt=hi - lo, k=4,
                                        n_lev=L, src_pitch=stacks[0].pitch, out_pitch=outs[0].pitch, layout=COLUMNS)
                return
            for s, o in zip(stacks, outs):
                native.regrid_ell(s.data, o.data, idx_d, w_d, n_src=n_src, n_tgt=hi - lo, k=4, n_lev=L, src_pitch=s.pitch, out_pitch=o.pitch, layout=COLUMNS)
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / 10 * 1e3)
        del outs
    t = np.array(times)
    print(f"{name:17s} world={world}: per-rank step ms {np.round(t, 3).tolist()}  max {t.max():.3f}  mean {t.mean():.3f}  "
          f"efficiency bound mean/max = {t.mean() / t.max():.3f}", flush=True)
