#!/usr/bin/env python3
"""What one step of the 8-rank strong-scaling line costs on its GPU beyond the kernel: K back-to-back launches of rank r's shard of the
headline (no events, one synchronise at the end: the bench's timed region) against the kernel's own duration (HIP events around single
launches include the launch floor; the rocprofv3 trace has the kernel alone), through the Python wrapper, through a bound call
(native.BoundCall: arguments converted once) and replayed from a HIP graph of K launches.

    python tools/experiments/strong_step_gaps.py [world] [rank]"""
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

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
K = 200
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
src, tgt = lookup('o1280'), lookup('0.25')
n_src, n_tgt, L = len(src['latitudes']), len(tgt['latitudes']), 137
idx, w = interp.knn_inverse_distance(src, tgt, k=4)
plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
b = plan.bounds(world, target_cost=TARGET_COST_SHORT_LAUNCH)
lo, hi = b[rank], b[rank + 1]
x = bench.synth_stack(src, L, torch.float64, dev, 0, COLUMNS)
out = Stack.empty(hi - lo, L, torch.float64, dev, COLUMNS)
i_d, w_d = torch.from_numpy(idx[lo:hi].astype(np.int32)).to(dev), torch.from_numpy(w[lo:hi]).to(dev)
kw = dict(n_src=n_src, n_tgt=hi - lo, k=4, n_lev=L, src_pitch=x.pitch, out_pitch=out.pitch, layout=COLUMNS)
plain = lambda: native.regrid_ell(x.data, out.data, i_d, w_d, **kw)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    bound = native.bind_regrid_ell(x.data, out.data, i_d, w_d, **kw)

def wall(fn, k=K):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3

ev = bench.time_launches(plain, 50, 5)
print(f"world {world} rank {rank}: {hi - lo} targets; HIP events around single launches: avg {ev[0]:.4f} ms, min {ev[1]:.4f} ms")
print(f"  {K} back-to-back launches, python wrapper: {wall(plain):.4f} ms per step")
with torch.cuda.stream(side):
    print(f"  {K} back-to-back launches, bound call:     {wall(bound):.4f} ms per step")
    # a HIP graph of K launches, replayed
    g = torch.cuda.CUDAGraph()
    bound()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        for _ in range(K): bound()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    print(f"  HIP graph of {K} launches, replayed:        {(time.perf_counter() - t0) / 5 / K * 1e3:.4f} ms per step")
