#!/usr/bin/env python3
"""Workload for the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, one counter set per run).

Runs, on one GPU, with the bench's O1280 -> 0.25 degree k=4 x137 inputs:
  1. a CALIBRATION launch with a known byte count in the same access width as the
     regrid kernel (16 B per lane): `atx_stream_copy` over n_src*n_lev*B bytes (reads them, writes the same);
  2. `--launches` regrid launches (`regrid_cols_ell_direct_kernel`).
The counters of (1) give the correction factor MI355X_MICROARCH.md §HBM asks for
(FETCH_SIZE reads ~1/2 of a wide coalesced stream on gfx950); tools/pmc_summarize.py
applies it to (2) and writes profiles/traffic.json.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 tools/pmc_probe.py
"""

from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--levels", type=int, default=137)
    ap.add_argument("--layout", default="columns", choices=["columns", "fields"])
    ap.add_argument("--case", default="ell", choices=["ell", "csr34", "csr916", "box", "perlevel", "fieldspw", "pad916", "config4", "config5"],
                    help="ell: fixed k (--k); csr34 / csr916: general CSR with ragged rows of 3-4 / 9-16 entries; box: O1280 -> 1 degree box averages (~100 per row)")
    ap.add_argument("--ordered", action="store_true", help="ell only: visit the targets in column blocks (atx_regrid_ell_ordered)")
    ap.add_argument("--chunk", type=int, default=0, help="ell only: one launch per chunk of this many levels (the level-chunk-major traversal, emulated)")
    ap.add_argument("--shard", type=int, default=-1, help="config4 / config5: time this one of the 8 traffic-balanced target shards (-1: all targets); "
                                                          "ell with --world: this rank's shard of the strong-scaling step")
    ap.add_argument("--world", type=int, default=1, help="ell only: the headline launch of rank --shard (default 0) of a --world-rank run, cut as bench.py cuts it "
                                                         "(GatherPlan.bounds(world, TARGET_COST_SHORT_LAUNCH)); recorded under the bench's traffic key `... gpus=<world>`")
    ap.add_argument("--tall", action="store_true", help="config4: 4 stacks of 6 x 137 levels (the variables of a point share a column) instead of 24 x 137; "
                                                        "config5: one stack of 3 x 137 levels instead of 137")
    ap.add_argument("--plain", action="store_true", help="config5: the gather alone instead of the fused regrid | orog_to_z | convert launch")
    ap.add_argument("--meta", default=os.path.join(ROOT, "gpurun_out", "pmc_meta.json"))
    args = ap.parse_args()

    graft.load_package()
    from anemoi_transform_amd import native
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.interp import knn_inverse_distance
    from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tdtype = torch.float32 if args.dtype == "f32" else torch.float64
    npdt = np.float32 if args.dtype == "f32" else np.float64
    itemsize = 4 if args.dtype == "f32" else 8

    if args.case in ("config4", "config5"):
        return baseline_configs(args, native, dev, tdtype, npdt, itemsize)

    src_grid, tgt_grid = lookup("o1280"), lookup("0.25")
    n_src, n_tgt = len(src_grid["latitudes"]), len(tgt_grid["latitudes"])
    idx64, w64 = knn_inverse_distance(src_grid, tgt_grid, k=args.k)
    if args.world > 1:  # one rank's shard of the strong-scaling step (bench.py's headline at N > 1)
        assert args.case == "ell" and args.layout == "columns" and not args.ordered and args.chunk <= 0
        from anemoi_transform_amd.gather import TARGET_COST_SHORT_LAUNCH, GatherPlan

        cut = GatherPlan(n_src, n_tgt, index=idx64, weights=w64).bounds(args.world, target_cost=TARGET_COST_SHORT_LAUNCH)
        lo, hi = cut[max(args.shard, 0)], cut[max(args.shard, 0) + 1]
        idx64, w64, n_tgt = idx64[lo:hi], w64[lo:hi], hi - lo
    idx = torch.from_numpy(idx64.astype(np.int32)).to(dev)
    w = torch.from_numpy(w64.astype(npdt)).to(dev) if args.k > 1 else None
    src = bench.synth_stack(src_grid, args.levels, tdtype, dev, 0, COLUMNS)
    out = Stack.empty(n_tgt, args.levels, tdtype, dev, COLUMNS)
    regrid_src, regrid_out = src, out
    if args.layout == "fields":
        regrid_src = src.to_layout(FIELDS)
        regrid_out = Stack.empty(n_tgt, args.levels, tdtype, dev, FIELDS)
    torch.cuda.synchronize()

    # 1. calibration: the library's fixed streaming copy (atx_stream_copy: one 16-byte vector per lane, aligned, known bytes) —
    #    the regrid kernel's access width.  The column-stack per-point kernel is launched too and reported with the same
    #    correction (it once showed a few % of boundary-line refetch that way).
    flat_src = torch.zeros(n_src * args.levels, dtype=tdtype, device=dev)
    flat_dst = torch.empty_like(flat_src)
    native.stream_copy(flat_src, flat_dst)
    torch.cuda.synchronize()
    calib_bytes = n_src * args.levels * itemsize
    del flat_src, flat_dst
    prog = native.level_program([[(native.OP_COPY, 0, 0.0, 0.0)] * args.levels], dev)
    copy = src.new_like()
    native.pointwise_stack(src.data, copy.data, n_pts=n_src, n_lev=args.levels, x_pitch=src.pitch, y_pitch=copy.pitch,
                           layout=COLUMNS, prog=prog, n_stage=1)
    torch.cuda.synchronize()
    covered = (args.levels + (16 // itemsize) - 1) // (16 // itemsize) * (16 // itemsize)
    cols_copy_bytes = n_src * covered * itemsize

    # 2. the regrid launches
    from anemoi_transform_amd import interp

    kernel = "regrid_fields_ell_kernel" if args.layout == "fields" else ("regrid_cols_ell_direct_kernel" if args.k <= 4 else "regrid_cols_ell_kernel")
    alg = bench.algorithmic_bytes(args.levels, itemsize, int(np.unique(idx64).size), n_tgt, args.k)
    config = f"o1280->0.25 k={args.k} L={args.levels} {args.dtype} {args.layout} gpus={args.world}" + (f" rank {args.shard}" if args.world > 1 and args.shard > 0 else "")
    if args.case == "ell":
        per16 = 16 // itemsize
        chunk = args.levels if args.chunk <= 0 else (args.chunk + per16 - 1) // per16 * per16
        cuts = list(range(0, args.levels, chunk)) + [args.levels]
        if len(cuts) > 2:
            config += f" level-chunks={chunk}"
        rows = None
        if args.ordered:
            from anemoi_transform_amd.gather import column_block_order

            order = column_block_order(tgt_grid["latitudes"], tgt_grid["longitudes"])
            idx = torch.from_numpy(np.ascontiguousarray(idx64[order]).astype(np.int32)).to(dev)
            w = torch.from_numpy(np.ascontiguousarray(w64[order]).astype(npdt)).to(dev) if args.k > 1 else None
            rows = torch.from_numpy(order).to(dev)
            config += " column-block order"
        for _ in range(args.launches):
            for a, b in zip(cuts[:-1], cuts[1:]):
                s_, o_ = (regrid_src.data[:, a:], regrid_out.data[:, a:]) if len(cuts) > 2 else (regrid_src.data, regrid_out.data)
                native.regrid_ell(s_, o_, idx, w, n_src=n_src, n_tgt=n_tgt, k=args.k, n_lev=b - a,
                                  src_pitch=regrid_src.pitch, out_pitch=regrid_out.pitch, layout=regrid_src.layout, tgt_rows=rows)
    elif args.case == "pad916":  # ragged rows of 9-16 entries the way regrid(matrix=...) runs them: padded to 16, targets in column blocks
        from anemoi_transform_amd.gather import GatherPlan, target_order_for

        i64, w64k = knn_inverse_distance(src_grid, tgt_grid, k=16, device=True, ties="index")
        keep = np.random.default_rng(16).random(i64.shape) < 0.75
        keep[:, :9] = True
        indptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))])
        plan = GatherPlan.from_matrix(dict(matrix_data=w64k[keep], matrix_indices=i64[keep], matrix_indptr=indptr, matrix_shape=(n_tgt, n_src)))
        assert plan.padded and plan.k == 16
        if args.ordered:
            plan.order_targets(target_order_for(tgt_grid["latitudes"], tgt_grid["longitudes"], 16))
        kernel = "regrid_cols_ell_direct_kernel"
        alg = args.levels * itemsize * (int(np.unique(i64[keep]).size) + n_tgt) + int(keep.sum()) * (4 + itemsize) + 4 * n_tgt
        config = f"o1280 rows of 9-16 padded to 16 L={args.levels} {args.dtype} columns gpus=1" + (" column-block order" if args.ordered else "")
        for _ in range(args.launches):
            plan.apply(src)
    elif args.case in ("perlevel", "fieldspw"):  # per-point kernels of round 3 (reported under the "regrid" keys of the summary)
        L = args.levels
        if args.case == "perlevel":  # a scale per level on a column stack: pointwise_cols_levels_kernel
            kernel, x_in, layout = "pointwise_cols_levels_kernel", src, COLUMNS
        else:  # the same program on a field-major stack: pointwise_fields_rows_kernel
            kernel, x_in, layout = "pointwise_fields_rows_kernel", src.to_layout(FIELDS), FIELDS
        y_out = x_in.new_like()
        per = native.level_program([[(native.OP_AFFINE, 0, 1.0 + 0.001 * l, -273.15) for l in range(L)]], dev)
        alg = 2 * n_src * L * itemsize
        config = f"o1280 {args.case} L={L} {args.dtype} gpus=1"
        for _ in range(args.launches):
            native.pointwise_stack(x_in.data, y_out.data, n_pts=n_src, n_lev=L, x_pitch=x_in.pitch, y_pitch=y_out.pitch, layout=layout, prog=per,
                                   n_stage=1)
    else:
        kernel = "regrid_cols_csr_kernel"
        if args.case == "box":  # every 1-degree cell averages the O1280 points inside it: each source column is read exactly once
            one = lookup([1.0, 1.0])
            n_rows = len(one["latitudes"])
            cell = (np.rint(90.0 - src_grid["latitudes"]).astype(np.int64) * 360 + np.mod(np.rint(src_grid["longitudes"]).astype(np.int64), 360))
            order = np.argsort(cell, kind="stable")
            counts = np.bincount(cell, minlength=n_rows)
            indptr = np.concatenate([[0], np.cumsum(counts)])
            data, indices = (1.0 / np.maximum(counts, 1))[cell[order]], order
        else:
            kk = 4 if args.case == "csr34" else 16
            i64, w64k = knn_inverse_distance(src_grid, tgt_grid, k=kk, device=kk > 4, **({"ties": "index"} if kk > 4 else {}))
            if args.case == "csr34":
                keep = (np.arange(i64.size) % 9 != 0).reshape(i64.shape)
            else:
                keep = np.random.default_rng(16).random(i64.shape) < 0.75
                keep[:, :9] = True
            n_rows = n_tgt
            indptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))])
            data, indices = w64k[keep], i64[keep]
        nnz = int(len(indices))
        alg = args.levels * itemsize * (int(np.unique(indices).size) + n_rows) + nnz * (4 + itemsize) + 4 * n_rows
        config = f"o1280 {args.case} L={args.levels} {args.dtype} columns gpus=1"
        csr_out = Stack.empty(n_rows, args.levels, tdtype, dev, COLUMNS)
        d_ptr, d_idx, d_w = (torch.from_numpy(indptr.astype(np.int32)).to(dev), torch.from_numpy(indices.astype(np.int32)).to(dev),
                             torch.from_numpy(data.astype(npdt)).to(dev))
        for _ in range(args.launches):
            native.regrid_csr(src.data, csr_out.data, d_ptr, d_idx, d_w, n_src=n_src, n_tgt=n_rows, nnz=nnz, n_lev=args.levels,
                              src_pitch=src.pitch, out_pitch=csr_out.pitch, layout=COLUMNS)
    torch.cuda.synchronize()

    meta = {
        "config": config,
        "calibration_kernel": "stream_copy_kernel",
        "cols_copy_bytes": cols_copy_bytes,
        "calibration_read_bytes": calib_bytes,
        "calibration_write_bytes": calib_bytes,
        "regrid_kernel": kernel,
        "regrid_launches": args.launches,
        "algorithmic_bytes_per_launch": alg,
    }
    os.makedirs(os.path.dirname(args.meta), exist_ok=True)
    json.dump(meta, open(args.meta, "w"), indent=1)
    print(json.dumps(meta))


def baseline_configs(args, native, dev, tdtype, npdt, itemsize):
    """BASELINE configs[3] and [4] as bench.py times them (extras.config4 / extras.config5), under the counters:
    config4 — O1280 -> N320-sized, k = 4, 3 288 fields resident as 24 stacks of 137 levels (or, --tall, 4 stacks of 822), one of the 8
    target shards (--shard) or all targets, one batched step per repetition; config5 — O2560 -> 0.25 degree, 137 levels (or, --tall,
    3 x 137 sharing a column), the fused regrid | orog_to_z | convert launch (or, --plain, the gather alone)."""
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.interp import knn_inverse_distance
    from anemoi_transform_amd.stack import COLUMNS, Stack

    def calibrate(n_elems):  # the library's fixed streaming copy over a known number of bytes, 16 bytes per lane
        a = torch.zeros(n_elems, dtype=tdtype, device=dev)
        b = torch.empty_like(a)
        native.stream_copy(a, b)
        torch.cuda.synchronize()
        return n_elems * itemsize

    which = "all targets" if args.shard < 0 else f"shard {args.shard} of 8"
    if args.case == "config4":
        g_src, g_tgt = lookup("o1280"), lookup("n320-sized")
        n_src, n_tgt, n_var, n_time = len(g_src["latitudes"]), len(g_tgt["latitudes"]), 6, 4
        idx, w = knn_inverse_distance(g_src, g_tgt, k=4)
        cuts = GatherPlan(n_src, n_tgt, index=idx, weights=w).bounds(8)
        lo, hi = (0, n_tgt) if args.shard < 0 else (cuts[args.shard], cuts[args.shard + 1])
        n_stack, n_lev = (n_time, n_var * args.levels) if args.tall else (n_var * n_time, args.levels)
        calib = calibrate(n_src * args.levels)
        gen = torch.Generator(device=dev)
        srcs = []
        for i in range(n_stack):
            gen.manual_seed(bench.SEED + 7 * i)
            st = Stack.empty(n_src, n_lev, tdtype, dev, COLUMNS, zero=True)
            st.data[:, :n_lev].normal_(250.0 + 5.0 * (i // 4), 20.0, generator=gen)
            srcs.append(st)
        outs = [Stack.empty(hi - lo, n_lev, tdtype, dev, COLUMNS) for _ in range(n_stack)]
        idx_d, w_d, rows_d = bench.ordered_tables(idx, w, g_tgt, lo, hi, npdt, dev)
        torch.cuda.synchronize()
        for _ in range(args.launches):
            native.regrid_ell_batch([s_.data for s_ in srcs], [o.data for o in outs], idx_d, w_d, n_src=n_src, n_tgt=hi - lo, k=4, n_lev=n_lev,
                                    src_pitch=srcs[0].pitch, out_pitch=outs[0].pitch, layout=COLUMNS, tgt_rows=rows_d)
        alg = n_stack * bench.algorithmic_bytes(n_lev, itemsize, int(np.unique(idx[lo:hi]).size), hi - lo, 4)
        config = f"config4 o1280->n320-sized k=4 {n_stack}x{n_lev} levels {args.dtype} columns {which}"
    else:
        class A:  # what bench.config5_case reads of the bench's arguments
            levels, natural_order = args.levels, False
        calib = calibrate(len(lookup("o1280")["latitudes"]) * args.levels)
        rank, world = (0, 1) if args.shard < 0 else (args.shard, 8)
        plain, fused, _, _, alg, keep = bench.config5_case(A, dev, tdtype, npdt, rank=rank, world=world, variables=3 if args.tall else 1)
        torch.cuda.synchronize()
        for _ in range(args.launches):
            (plain if args.plain else fused)()
        n_lev = args.levels * (3 if args.tall else 1)
        config = (f"config5 o2560->0.25 k=4 L={n_lev} {args.dtype} columns {'regrid only' if args.plain else 'fused regrid|orog_to_z|convert'} {which}")
    torch.cuda.synchronize()
    meta = {"config": config, "calibration_kernel": "stream_copy_kernel", "calibration_read_bytes": calib, "calibration_write_bytes": calib,
            "regrid_kernel": "regrid_cols_ell_direct_kernel", "regrid_launches": args.launches, "algorithmic_bytes_per_launch": alg}
    os.makedirs(os.path.dirname(args.meta), exist_ok=True)
    json.dump(meta, open(args.meta, "w"), indent=1)
    print(json.dumps(meta))


if __name__ == "__main__":
    main()
