#!/usr/bin/env python3
"""Headline benchmark: grid-points/s regridded, O1280 -> 0.25 degree x 137 levels.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one pass of the regrid hot path over one batch of synthetic input
already resident in HBM: at N = 1 one 137-level stack on the O1280 octahedral
grid (6 599 680 points) is interpolated to the 0.25 degree lat-lon grid
(1 038 240 points) with k = 4 inverse-distance weights from cKDTree (BASELINE.json
configs[2], the config the metric is quoted on), float32, one `atx_regrid_ell`
launch.  At N > 1 (weak scaling, one process per GPU) the job is N such stacks
(N variables x 137 levels); the source stacks are exchanged ONCE by RCCL
broadcasts before the timed region (reported as `source_exchange_ms`), and the
target points are sharded N ways: every rank interpolates its 1/N slice of the
target grid for all N stacks — N launches per step, no collective in the data
path.  value = point-fields all ranks produced / max-over-ranks wall time.

Also reported on the same JSON line:
  roofline     algorithmic bytes of one launch / its average HIP-event duration,
               against the 8 TB/s HBM3E peak
  cpu_baseline the oracle's scipy `csr_array @ x` statement (the reference's CPU
               path, R: filters/fields/regrid.py:310) timed on one host core on a
               bounded sample of the same workload (rank 0, N = 1 only)
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
SEED = 20260630


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--src-grid", default="o1280")
    ap.add_argument("--tgt-grid", default="0.25")
    ap.add_argument("--levels", type=int, default=137)
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--layout", default="columns", choices=["columns", "fields"])
    ap.add_argument("--tile", type=int, default=0, help="targets per workgroup (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary k=1 / f64 / field-major lines")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="collective backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--exchange", default="broadcast", choices=["broadcast", "bands"],
                    help="N > 1 source exchange before timing: whole stacks by RCCL broadcast, or only the band of source columns "
                         "each rank's target slice references by send/recv (distributed.exchange_source_bands)")
    ap.add_argument("--end-to-end", action="store_true",
                    help="N > 1: also time one step INCLUDING the source exchange, broadcasts double-buffered against the launches "
                         "(distributed.pipelined_sharded_regrid); reported as `end_to_end`, never as `value`")
    ap.add_argument("--share-device", action="store_true",
                    help="REHEARSAL ONLY: all ranks use cuda:0 (exercises the N > 1 code path on a 1-GPU box; needs --backend gloo)")
    return ap.parse_args()


def synth_stack(grid, n_lev, dtype, dev, stack_id, layout):
    """v[l, p] = 280 + 30 sin(lat) cos(2 lon + 0.1 l) + N(0, 1)   (SURVEY.md §8d), built in HBM."""
    from anemoi_transform_amd.stack import COLUMNS, Stack

    n_pts = len(grid["latitudes"])
    lat = torch.from_numpy(np.deg2rad(grid["latitudes"])).to(dev)
    lon = torch.from_numpy(np.deg2rad(grid["longitudes"])).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(SEED + 1000 * stack_id)
    st = Stack.empty(n_pts, n_lev, dtype, dev, layout, zero=True)
    sin_lat = torch.sin(lat)
    for l in range(n_lev):
        v = 280.0 + 30.0 * sin_lat * torch.cos(2.0 * lon + 0.1 * l + 0.37 * stack_id)
        v = v + torch.randn(n_pts, dtype=torch.float64, device=dev, generator=gen)
        if layout == COLUMNS:
            st.data[:, l] = v.to(dtype)
        else:
            st.data[l, :n_pts] = v.to(dtype)
    return st


def algorithmic_bytes(n_lev, itemsize, n_unique, n_tgt, k):
    """L*B*(U + Nt) + Nt*k*4 + [k>1]*Nt*k*B."""
    b = n_lev * itemsize * (n_unique + n_tgt) + n_tgt * k * 4
    if k > 1:
        b += n_tgt * k * itemsize
    return b


def time_launches(fn, steps, warmup):
    """Average HIP-event duration (ms) of `fn` (one launch) on the current stream."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in evs]
    return float(np.mean(ms)), float(np.min(ms))


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 through torch.distributed.run (one process per GPU)")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    graft.load_package()
    from anemoi_transform_amd import native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.interp import knn_inverse_distance
    from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack

    native.load()  # fails loudly if the HIP extension is missing
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", 0 if args.share_device else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            assert not args.share_device, "RCCL needs one GPU per rank"
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group("gloo")

    tdtype = torch.float32 if args.dtype == "f32" else torch.float64
    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    itemsize = 4 if args.dtype == "f32" else 8
    layout = COLUMNS if args.layout == "columns" else FIELDS
    if args.tile:
        native.set_tuning(args.tile)

    # ---- one-off precompute on the host (not timed): grids, cKDTree indices + weights
    t0 = time.perf_counter()
    src_grid, tgt_grid = lookup(args.src_grid), lookup(args.tgt_grid)
    n_src, n_tgt = len(src_grid["latitudes"]), len(tgt_grid["latitudes"])
    idx64, w64 = knn_inverse_distance(src_grid, tgt_grid, k=args.k)
    precompute_s = time.perf_counter() - t0
    n_unique = int(np.unique(idx64).size)

    # target-point shard of this rank: contiguous, balanced by HBM traffic (GatherPlan.bounds) — equal-count
    # shards of a lat-lon target are 1.8x apart in cost (polar targets share their source columns)
    from anemoi_transform_amd.gather import GatherPlan

    plan = GatherPlan(n_src, n_tgt, index=idx64, weights=w64)
    bounds = plan.bounds(world)
    lo, hi = bounds[rank], bounds[rank + 1]
    # ---- sources resident in HBM before the timed region
    mine = synth_stack(src_grid, args.levels, tdtype, dev, rank, layout)
    stacks = [mine]
    exchange_ms = None
    band_lo, n_src_local = 0, n_src
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        if args.exchange == "bands":
            from anemoi_transform_amd.distributed import exchange_source_bands, source_band

            stacks, _ = exchange_source_bands(mine, plan)
            band_lo, band_hi = source_band(plan.shard(rank, world))
            n_src_local = band_hi - band_lo
            del mine
        else:
            stacks = []
            for r in range(world):
                buf = mine if r == rank else Stack.empty(n_src, args.levels, tdtype, dev, layout)
                dist.broadcast(buf.data, src=r)  # the one-off source exchange (RCCL)
                stacks.append(buf)
        torch.cuda.synchronize()
        dist.barrier()
        exchange_ms = (time.perf_counter() - t0) * 1e3
    idx_d = torch.from_numpy((idx64[lo:hi] - band_lo).astype(np.int32)).to(dev)
    w_d = torch.from_numpy(w64[lo:hi].astype(np_dtype)).to(dev)
    assert native.check_indices(idx_d, n_src_local) == 0
    outs = [Stack.empty(hi - lo, args.levels, tdtype, dev, layout) for _ in stacks]
    weighted = args.k > 1

    def launch(src, out, idx=idx_d, w=w_d, k=args.k, n_t=hi - lo):
        native.regrid_ell(src.data, out.data, idx, w if weighted or k > 1 else None, n_src=n_src_local, n_tgt=n_t, k=k,
                          n_lev=src.n_lev, src_pitch=src.pitch, out_pitch=out.pitch, layout=src.layout)

    def step():  # one launch over all stacks of the step (atx_regrid_ell_batch: grid.y = stack)
        if len(stacks) == 1:
            launch(stacks[0], outs[0])
        else:
            native.regrid_ell_batch([s.data for s in stacks], [o.data for o in outs], idx_d, w_d, n_src=n_src_local, n_tgt=hi - lo,
                                    k=args.k, n_lev=args.levels, src_pitch=stacks[0].pitch, out_pitch=outs[0].pitch, layout=layout)

    # ---- timed region: W warm-up steps, then exactly K steps between barriers
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    units_per_step = n_tgt * args.levels * world  # all ranks together: N stacks x the full target grid (shards tile it)
    value = units_per_step * args.steps / elapsed

    # ---- roofline of the dominant kernel, HIP events around single launches
    # (N > 1, column stacks: the step IS one batched launch over this rank's shard of all N stacks)
    one_launch = layout == COLUMNS or len(stacks) == 1
    avg_ms, min_ms = time_launches(step if one_launch else (lambda: launch(stacks[0], outs[0])), min(max(args.steps, 10), 200), 2)
    shard_unique = int(np.unique(idx64[lo:hi]).size)
    alg = algorithmic_bytes(args.levels, itemsize, shard_unique, hi - lo, args.k) * (len(stacks) if one_launch else 1)
    achieved = alg / (avg_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath))
            key = f"{args.src_grid}->{args.tgt_grid} k={args.k} L={args.levels} {args.dtype} {args.layout} gpus={world}"
            traffic = rec.get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "grid-points/sec regridded, O1280->0.25deg x137 levels",
        "value": value,
        "unit": "grid-points/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"{args.src_grid.upper()} ({n_src} pts) -> {args.tgt_grid} deg lat-lon ({n_tgt} pts), "
                        f"k={args.k} inverse-distance regrid x {args.levels} levels per stack",
            "layout": args.layout,
            "stacks_per_step": world,
            "sharding": ("target points over ranks (contiguous, traffic-balanced); sources exchanged once before timing by "
                         + ("RCCL broadcast" if args.exchange == "broadcast" else "band-limited send/recv")) if world > 1 else "single GPU",
            "launches_per_step_per_gpu": 1 if layout == COLUMNS else world,
            **({"rehearsal": "ranks share one GPU over gloo; not a scaling measurement"} if args.share_device else {}),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "regrid_cols_ell_direct_kernel" if layout == COLUMNS else "regrid_fields_ell_kernel",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": alg,
            "avg_launch_ms": avg_ms,
            "min_launch_ms": min_ms,
            "distinct_source_points": shard_unique,
        },
        "precompute_s": precompute_s,
    }
    if world > 1 and args.exchange == "broadcast":
        # comparison point (axis 2): shard the FIELDS instead of the target points — every rank interpolates the
        # whole target grid of its own stack, nothing is exchanged at all.  Same units per step; reported next to `value`.
        try:
            full_idx = torch.from_numpy(idx64.astype(np.int32)).to(dev)
            full_w = torch.from_numpy(w64.astype(np_dtype)).to(dev)
            own, full_out = stacks[rank], Stack.empty(n_tgt, args.levels, tdtype, dev, layout)

            def own_step():
                launch(own, full_out, idx=full_idx, w=full_w, n_t=n_tgt)

            for _ in range(args.warmup):
                own_step()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                own_step()
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            result["field_axis_sharding"] = {"value": units_per_step * args.steps / float(t.item()), "unit": "grid-points/s",
                                             "ms_per_step": float(t.item()) / args.steps * 1e3,
                                             "note": "each rank regrids its own stack to the full target grid; no source exchange"}
            del full_idx, full_w, full_out
        except Exception as e:  # a comparison line must never take the bench line down
            result["field_axis_sharding"] = {"error": f"{type(e).__name__}: {e}"}
    if world > 1 and args.end_to_end and layout == COLUMNS:
        from anemoi_transform_amd.distributed import pipelined_sharded_regrid

        del outs
        torch.cuda.empty_cache()
        mine_again = synth_stack(src_grid, args.levels, tdtype, dev, rank, layout)
        pipelined_sharded_regrid(plan, mine_again)  # warm-up (communicator, allocations)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        pipelined_sharded_regrid(plan, mine_again)
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        result["end_to_end"] = {"ms_per_step": float(t.item()) * 1e3, "value": units_per_step / float(t.item()), "unit": "grid-points/s",
                                "note": "one step including the exchange of the N source stacks (broadcast r+1 overlapped with launch r)"}
    if exchange_ms is not None:
        result["source_exchange_ms"] = exchange_ms
        result["source_exchange"] = args.exchange
        result["source_bytes_held_per_gpu"] = sum(s.data.numel() * s.data.element_size() for s in stacks)

    if rank == 0 and world == 1:
        # ---- parity spot check + CPU baseline on a bounded sample of the same workload
        sys.path.insert(0, ROOT)
        from oracle import oracle  # checker / baseline only

        n_sample = min(8, args.levels)
        levels = np.linspace(0, args.levels - 1, n_sample).astype(int)
        sample64 = np.stack([stacks[0].level_numpy(int(l)).astype(np.float64) for l in levels])
        got = np.stack([outs[0].level_numpy(int(l)) for l in levels])
        indptr = (np.arange(n_tgt + 1, dtype=np.int64) * args.k).astype(np.int32)
        indices = idx64.astype(np.int32).reshape(-1)
        want = np.stack([
            oracle.csr_apply(w64.astype(np_dtype).reshape(-1), indices, indptr, (n_tgt, n_src), f.astype(np_dtype))
            for f in sample64
        ])
        result["parity_max_rel_err"] = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-30)))

        if not args.no_cpu_baseline:
            torch.set_num_threads(1)
            data64 = w64.reshape(-1)
            # the reference's dtype is float64 (to_numpy default): time csr_array @ x per field, one thread
            from scipy.sparse import csr_array

            matrix = csr_array((data64, indices, indptr), shape=(n_tgt, n_src))
            matrix @ sample64[0]
            n_done, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < args.cpu_seconds:
                for f in sample64:
                    _ = matrix @ f
                    n_done += 1
            cpu_s = time.perf_counter() - t0
            result["cpu_baseline"] = {
                "value": n_done * n_tgt / cpu_s,
                "unit": "grid-points/s",
                "cores": 1,
                "kind": "port",
                "sample": f"{n_done} fields ({n_sample} distinct levels of the same synthetic stack, float64 as in the "
                          f"reference) x scipy csr_array(k={args.k}) @ x, {cpu_s:.1f} s on 1 thread; "
                          f"host has {os.cpu_count()} logical cores",
                "ms_per_field": cpu_s / n_done * 1e3,
            }

        if not args.no_extras:
            extras = {}
            # k = 1 nearest-neighbour gather (R: regrid.py:380), same stack
            idx1 = torch.from_numpy(idx64[:, 0].astype(np.int32).copy()).to(dev)
            ms1, _ = time_launches(lambda: launch(stacks[0], outs[0], idx=idx1, w=None, k=1, n_t=n_tgt), 10, 2)
            alg1 = algorithmic_bytes(args.levels, itemsize, int(np.unique(idx64[:, 0]).size), n_tgt, 1)
            extras["nearest_k1"] = {"value": n_tgt * args.levels / (ms1 * 1e-3), "avg_launch_ms": ms1,
                                    "achieved_GBs": alg1 / (ms1 * 1e-3) / 1e9, "frac": alg1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # config-5 shape on the same stack: regrid -> orog_to_z -> convert fused in ONE launch
            prog = native.level_program([[(native.OP_MUL, 0, 9.80665, 0.0)] * args.levels,
                                         [(native.OP_AFFINE, 0, 1.0, -273.15)] * args.levels], dev)
            msf, _ = time_launches(lambda: native.regrid_ell(
                stacks[0].data, outs[0].data, idx_d, w_d, n_src=n_src, n_tgt=n_tgt, k=args.k, n_lev=args.levels,
                src_pitch=stacks[0].pitch, out_pitch=outs[0].pitch, layout=stacks[0].layout, prog=prog, n_stage=2), 10, 2)
            extras["fused_regrid_orog_to_z_convert"] = {"value": n_tgt * args.levels / (msf * 1e-3), "avg_launch_ms": msf,
                                                        "achieved_GBs": alg / (msf * 1e-3) / 1e9,
                                                        "frac": alg / (msf * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # the one-off index build on the device instead of cKDTree (interp.nearest_grid_points_device)
            from anemoi_transform_amd import interp as _interp

            torch.cuda.synchronize()
            t0 = time.perf_counter()
            di, dd = _interp.nearest_grid_points_device(src_grid["latitudes"], src_grid["longitudes"], tgt_grid["latitudes"],
                                                        tgt_grid["longitudes"], num_neighbours_to_return=args.k, return_distances=True)
            extras["knn_device_s"] = time.perf_counter() - t0
            extras["knn_rows_identical_to_ckdtree"] = float((di.reshape(n_tgt, -1) == idx64).all(axis=1).mean())
            del stacks, outs, mine
            torch.cuda.empty_cache()
            for name, dt, npdt, isz, lay in (("f64_columns", torch.float64, np.float64, 8, COLUMNS),
                                             ("f32_fields", torch.float32, np.float32, 4, FIELDS)):
                s = synth_stack(src_grid, args.levels, dt, dev, 0, lay)
                o = Stack.empty(n_tgt, args.levels, dt, dev, lay)
                wd = torch.from_numpy(w64.astype(npdt)).to(dev)
                ms, _ = time_launches(lambda: launch(s, o, w=wd), 10, 2)
                a = algorithmic_bytes(args.levels, isz, n_unique, n_tgt, args.k)
                extras[name] = {"value": n_tgt * args.levels / (ms * 1e-3), "avg_launch_ms": ms,
                                "achieved_GBs": a / (ms * 1e-3) / 1e9, "frac": a / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                del s, o, wd
                torch.cuda.empty_cache()
            # the CPU's best case (SURVEY.md §8d baseline B): the same statement on 16 worker processes, run as a child
            # process that never touches the GPU; reported next to the one-thread "as the reference runs" baseline
            if not args.no_cpu_baseline:
                import subprocess

                try:
                    child = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_all_cores.py"), "--seconds", "8",
                                            "--src-grid", args.src_grid, "--tgt-grid", args.tgt_grid, "--k", str(args.k)],
                                           capture_output=True, text=True, timeout=180)
                    extras["cpu_all_cores"] = json.loads(child.stdout.strip().splitlines()[-1])
                except Exception as e:  # a baseline must never take the bench line down
                    extras["cpu_all_cores"] = {"error": f"{type(e).__name__}: {e}"}
            result["extras"] = extras

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
