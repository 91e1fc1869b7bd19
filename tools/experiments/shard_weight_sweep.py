#!/usr/bin/env python3
"""Which weight of the shard cost model (gather.TARGET_COST: cost of a target = the source columns it is the first to read + weight)
gives the fastest SLOWEST shard — for the strong-scaling step (ONE 137-level stack per launch: the bench's headline at N > 1) and for
the weak-scaling step (N stacks in one batched launch), float64 and float32, N = 8 and 4.  Every shard timed alone on one MI355X.

    python tools/experiments/shard_weight_sweep.py [weights ...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft, bench
graft.load_package()
from anemoi_transform_amd import native, interp, gather
from anemoi_transform_amd.gather import GatherPlan
from anemoi_transform_amd.grids import lookup
from anemoi_transform_amd.stack import COLUMNS, Stack

weights = [float(a) for a in sys.argv[1:]] or [0.6, 0.9, 1.1, 1.3, 1.5, 1.8]
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
src, tgt = lookup('o1280'), lookup('0.25')
n_src, n_tgt, L = len(src['latitudes']), len(tgt['latitudes']), 137
idx, w = interp.knn_inverse_distance(src, tgt, k=4)
plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
for tdt, npdt, tag in ((torch.float64, np.float64, 'f64'), (torch.float32, np.float32, 'f32')):
    stacks = [bench.synth_stack(src, L, tdt, dev, s, COLUMNS) for s in range(8)]
    full = Stack.empty(n_tgt, L, tdt, dev, COLUMNS)
    idx_full, w_full = torch.from_numpy(idx.astype(np.int32)).to(dev), torch.from_numpy(w.astype(npdt)).to(dev)
    n1 = bench.time_launches(lambda: native.regrid_ell(stacks[0].data, full.data, idx_full, w_full, n_src=n_src, n_tgt=n_tgt, k=4, n_lev=L, src_pitch=stacks[0].pitch,
                                                       out_pitch=full.pitch, layout=COLUMNS), 20, 3)[0]
    print(f"{tag}: N = 1 launch {n1:.4f} ms", flush=True)
    for world in (8, 4):
        for wgt in weights:
            b = plan.bounds(world, target_cost=wgt)
            single, batched = [], []
            for r in range(world):
                lo, hi = b[r], b[r + 1]
                i_d, w_d = torch.from_numpy(idx[lo:hi].astype(np.int32)).to(dev), torch.from_numpy(w[lo:hi].astype(npdt)).to(dev)
                outs = [Stack.empty(hi - lo, L, tdt, dev, COLUMNS) for _ in range(world)]
                single.append(bench.time_launches(lambda: native.regrid_ell(stacks[0].data, outs[0].data, i_d, w_d, n_src=n_src, n_tgt=hi - lo, k=4, n_lev=L,
                                                                            src_pitch=stacks[0].pitch, out_pitch=outs[0].pitch, layout=COLUMNS), 20, 3)[0])
                batched.append(bench.time_launches(lambda: native.regrid_ell_batch([s.data for s in stacks[:world]], [o.data for o in outs], i_d, w_d, n_src=n_src,
                                                                                   n_tgt=hi - lo, k=4, n_lev=L, src_pitch=stacks[0].pitch, out_pitch=outs[0].pitch,
                                                                                   layout=COLUMNS), 10, 2)[0])
                del outs, i_d, w_d
            s_, b_ = np.array(single), np.array(batched)
            print(f"{tag} world={world} weight={wgt:.1f}: strong step (1 stack) slowest {s_.max():.4f} ms -> speed-up bound {n1 / s_.max():.2f} "
                  f"[{' '.join(f'{x:.3f}' for x in s_)}] | weak step ({world} stacks) slowest {b_.max():.4f} ms = {b_.max() / n1:.3f} x N=1 "
                  f"[{' '.join(f'{x:.3f}' for x in b_)}]", flush=True)
    del stacks, full
    torch.cuda.empty_cache()
