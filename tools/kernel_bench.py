#!/usr/bin/env python3
"""Per-kernel timing of libatx at BASELINE sizes (O1280 stack of 137 levels), with the HBM roofline
fraction of each on its ALGORITHMIC bytes.  One JSON document on stdout / --out.

    python tools/kernel_bench.py --out gpurun_out/kernels.json
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
import bench  # noqa: E402

PEAK = 8e12


def timeit(fn, n=10, warm=2):
    """Median of n per-call HIP-event durations: one call in a few hundred stalls for tens of ms on a shared host (a lone 4.4 ms
    reading among 1.9 ms ones, `profiles/r03_padded_order_probe.log`), which a mean over 10 carries into the table."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--levels", type=int, default=137)
    args = ap.parse_args()
    graft.load_package()
    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.gather import GatherPlan
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    L = args.levels
    src_grid, tgt_grid = lookup("o1280"), lookup("0.25")
    n_src, n_tgt = len(src_grid["latitudes"]), len(tgt_grid["latitudes"])
    idx, w = interp.knn_inverse_distance(src_grid, tgt_grid, k=4)
    res = {}

    def record(name, ms, alg_bytes, note=""):
        res[name] = {"ms": ms, "algorithmic_bytes": alg_bytes, "GBs": alg_bytes / ms / 1e6, "frac_of_8TBs": alg_bytes / (ms * 1e-3) / PEAK,
                     "note": note}
        print(f"{name:42s} {ms:9.4f} ms  {alg_bytes / ms / 1e6:9.1f} GB/s  frac {alg_bytes / (ms * 1e-3) / PEAK:.3f}  {note}", flush=True)

    for tdt, npdt, B, tag in ((torch.float32, np.float32, 4, "f32"), (torch.float64, np.float64, 8, "f64")):
        x = bench.synth_stack(src_grid, L, tdt, dev, 0, COLUMNS)
        U4, U1 = int(np.unique(idx).size), int(np.unique(idx[:, 0]).size)
        # ---- regrid variants
        plan4 = GatherPlan(n_src, n_tgt, index=idx, weights=w)
        plan1 = GatherPlan(n_src, n_tgt, index=idx[:, 0])
        record(f"regrid_ell k=4 {tag} columns", timeit(lambda: plan4.apply(x)), bench.algorithmic_bytes(L, B, U4, n_tgt, 4), "incl. output allocation")
        record(f"regrid_ell k=1 {tag} columns", timeit(lambda: plan1.apply(x)), bench.algorithmic_bytes(L, B, U1, n_tgt, 1))
        keep = (np.arange(idx.size) % 9 != 0).reshape(idx.shape)
        indptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))])
        csr = GatherPlan(n_src, n_tgt, csr=(w[keep], idx[keep], indptr))
        nnz = int(keep.sum())
        csr_bytes = L * B * (int(np.unique(idx[keep]).size) + n_tgt) + nnz * (4 + B) + 4 * n_tgt
        record(f"regrid_csr ragged(3-4) {tag} columns", timeit(lambda: csr.apply(x)), csr_bytes, "general CSR kernel")
        padded = GatherPlan.from_matrix(dict(matrix_data=w[keep], matrix_indices=idx[keep], matrix_indptr=indptr, matrix_shape=(n_tgt, n_src)))
        assert padded.padded
        record(f"regrid ragged(3-4) as padded fixed-k {tag}", timeit(lambda: padded.apply(x)), csr_bytes, "what regrid(matrix=...) uses for short ragged rows")
        tile = 16 if B == 4 else 8  # what the tiled kernels' heuristic picks at 137 levels
        native.set_tuning(tile)
        record(f"regrid_csr ragged(3-4) {tag} columns, TILED kernel", timeit(lambda: csr.apply(x)), csr_bytes, "round 1's kernel (LDS-staged CSR slice)")
        native.set_tuning(0)
        # long rows: 16 nearest neighbours, and the same with entries dropped (rows of 9-16)
        idx16, w16 = interp.knn_inverse_distance(src_grid, tgt_grid, k=16, device=True, ties="index")
        U16 = int(np.unique(idx16).size)
        plan16 = GatherPlan(n_src, n_tgt, index=idx16, weights=w16)
        bytes16 = bench.algorithmic_bytes(L, B, U16, n_tgt, 16)
        record(f"regrid_ell k=16 {tag} columns", timeit(lambda: plan16.apply(x)), bytes16, "compile-time k = 16 on the direct kernel; targets in natural order, workgroups dealt to the XCDs in stripes (round 4)")
        from anemoi_transform_amd.gather import target_order_for

        plan16.order_targets(target_order_for(tgt_grid["latitudes"], tgt_grid["longitudes"], 16))
        record(f"regrid_ell k=16 {tag} columns, targets in column blocks", timeit(lambda: plan16.apply(x)), bytes16,
               "what regrid(matrix=...) does for k >= 5 on large output grids (atx_regrid_ell_ordered; same bits)")
        rng16 = np.random.default_rng(16)
        keep16 = rng16.random(idx16.shape) < 0.75
        keep16[:, :9] = True
        indptr16 = np.concatenate([[0], np.cumsum(keep16.sum(axis=1))])
        csr16 = GatherPlan(n_src, n_tgt, csr=(w16[keep16], idx16[keep16], indptr16))
        nnz16 = int(keep16.sum())
        csr16_bytes = L * B * (int(np.unique(idx16[keep16]).size) + n_tgt) + nnz16 * (4 + B) + 4 * n_tgt
        record(f"regrid_csr rows of 9-16 {tag} columns", timeit(lambda: csr16.apply(x)), csr16_bytes, "general CSR, direct kernel")
        native.set_tuning(tile)
        record(f"regrid_csr rows of 9-16 {tag} columns, TILED kernel", timeit(lambda: csr16.apply(x)), csr16_bytes, "round 1's kernel")
        native.set_tuning(0)
        padded16 = GatherPlan.from_matrix(dict(matrix_data=w16[keep16], matrix_indices=idx16[keep16], matrix_indptr=indptr16, matrix_shape=(n_tgt, n_src)))
        assert padded16.padded and padded16.k == 16
        record(f"regrid rows of 9-16 as padded fixed-k {tag}", timeit(lambda: padded16.apply(x)), csr16_bytes, "what regrid(matrix=...) uses for ragged rows up to 16 entries")
        padded16.order_targets(target_order_for(tgt_grid["latitudes"], tgt_grid["longitudes"], 16))
        record(f"regrid rows of 9-16 as padded fixed-k {tag}, targets in column blocks", timeit(lambda: padded16.apply(x)), csr16_bytes,
               "the same in the order the regrid filter's policy picks on large output grids")
        del padded16
        csr16.order_targets(target_order_for(tgt_grid["latitudes"], tgt_grid["longitudes"], 12))
        record(f"regrid_csr rows of 9-16 {tag} columns, targets in column blocks", timeit(lambda: csr16.apply(x)), csr16_bytes,
               "what regrid(matrix=...) does for long ragged rows on large output grids (atx_regrid_csr_ordered; same bits)")
        del plan16, csr16, idx16, w16
        # coarsening by box averages (a conservative-style matrix): every 1-degree cell averages the ~200 O1280 points inside it
        one = lookup([1.0, 1.0])
        n_one = len(one["latitudes"])
        cell = (np.rint(90.0 - src_grid["latitudes"]).astype(np.int64) * 360 + np.mod(np.rint(src_grid["longitudes"]).astype(np.int64), 360))
        order_b = np.argsort(cell, kind="stable")
        counts_b = np.bincount(cell, minlength=n_one)
        indptr_b = np.concatenate([[0], np.cumsum(counts_b)])
        data_b = (1.0 / np.maximum(counts_b, 1))[cell[order_b]]
        box = GatherPlan(n_src, n_one, csr=(data_b, order_b.astype(np.int32), indptr_b))
        box_bytes = L * B * (n_src + n_one) + n_src * (4 + B) + 4 * n_one
        record(f"regrid_csr box average O1280->1deg {tag} (rows of ~{int(counts_b.mean())})", timeit(lambda: box.apply(x)), box_bytes,
               "general CSR, every source column read once")
        del box
        prog = native.level_program([[(native.OP_MUL, 0, 9.80665, 0.0)] * L, [(native.OP_AFFINE, 0, 1.0, -273.15)] * L], dev)
        record(f"regrid_ell k=4 {tag} + 2-stage epilogue", timeit(lambda: plan4.apply(x, prog=prog, n_stage=2)),
               bench.algorithmic_bytes(L, B, U4, n_tgt, 4), "fused regrid -> orog_to_z -> rescale, every level")
        cp = (native.OP_COPY, 0, 0.0, 0.0)
        prog5 = native.level_program([[cp] * (L - 1) + [(native.OP_MUL, 0, 9.80665, 0.0)], [(native.OP_AFFINE, 0, 1.0, -273.15)] * (L - 1) + [cp]], dev)
        record(f"regrid_ell k=4 {tag} + config-5 epilogue", timeit(lambda: plan4.apply(x, prog=prog5, n_stage=2)),
               bench.algorithmic_bytes(L, B, U4, n_tgt, 4), "136 levels convert, 1 level orog_to_z: operators by value, two pieces")
        # ---- per-point
        y = x.new_like()
        record(f"(ceiling) torch copy_ of the stack {tag}", timeit(lambda: y.data.copy_(x.data)), 2 * x.data.numel() * B,
               "device-to-device copy of the same bytes: the practical read+write streaming rate")
        p1 = native.level_program([[(native.OP_AFFINE, 0, 1.0, -273.15)] * L], dev)
        kw = dict(n_pts=n_src, n_lev=L, x_pitch=x.pitch, y_pitch=y.pitch, layout=COLUMNS)
        stack_bytes = n_src * L * B
        record(f"pointwise affine {tag} out-of-place", timeit(lambda: native.pointwise_stack(x.data, y.data, prog=p1, n_stage=1, **kw)), 2 * stack_bytes)
        record(f"pointwise affine {tag} in-place", timeit(lambda: native.pointwise_stack(y.data, y.data, prog=p1, n_stage=1, **kw)), 2 * stack_bytes)
        p2 = native.level_program([[(native.OP_MUL, 0, 9.80665, 0.0)] * L, [(native.OP_AFFINE, 0, 1.0, -273.15)] * L], dev)
        record(f"pointwise 2 stages, uniform over the levels {tag} out-of-place", timeit(lambda: native.pointwise_stack(x.data, y.data, prog=p2, n_stage=2, **kw)), 2 * stack_bytes,
               "operators by value")
        pl = native.level_program([[(native.OP_AFFINE, 0, 1.0 + 0.001 * l, -273.15) for l in range(L)]], dev)
        record(f"pointwise affine, a different operator per level {tag} out-of-place", timeit(lambda: native.pointwise_stack(x.data, y.data, prog=pl, n_stage=1, **kw)), 2 * stack_bytes,
               "per-level tables: LDS kernel (f32 out of place), typed no-loop kernel (f64)")
        pe = native.level_program([[(native.OP_LOG, 0, 0.0, 0.0)] * L, [(native.OP_EXP, 0, 0.0, 0.0)] * L], dev)
        record(f"pointwise log then exp (sp_to_lnsp | lnsp_to_sp) {tag} out-of-place", timeit(lambda: native.pointwise_stack(x.data, y.data, prog=pe, n_stage=2, **kw)), 2 * stack_bytes,
               "log and exp (the library's own routines in float64, atx_common.hpp) on every element")
        pm = (torch.rand(n_src + 8, device=dev) < 0.3).to(torch.uint8)
        pmask = native.level_program([[(native.OP_COPY, 1, 0.0, 0.0)] * L], dev)
        record(f"apply_mask {tag}", timeit(lambda: native.pointwise_stack(x.data, y.data, prog=pmask, n_stage=1, point_mask=pm, **kw)), 2 * stack_bytes + n_src)
        one = native.level_program([[(native.OP_AFFINE, 0, 2.0, 1.0)] + [(native.OP_COPY, 0, 0.0, 0.0)] * (L - 1)], dev)
        record(f"pointwise 1 of {L} levels selected {tag} in-place", timeit(lambda: native.pointwise_stack(y.data, y.data, prog=one, n_stage=1, **kw)),
               2 * n_src * B, "untouched levels are skipped; 16-B vector granularity")
        record(f"reduce min+max of the stack {tag} (one pass)", timeit(lambda: native.reduce_stack(x.data, native.RED_MINMAX, n_pts=n_src, n_lev=L, pitch=x.pitch, layout=COLUMNS)),
               stack_bytes, "the range check of cos_sin_from_rad; includes the device->host read (round 1: two passes of 0.79 ms)")
        # ---- multi-input
        z = x.new_like()
        # snow depth (m of water equivalent) and density as they occur — in REGIONS, as on a real field (points are stored by
        # latitude): ~55 % of the points bare (sd = 0), ~35 % deep snow, ~10 % a thin cover where tanh really has to be evaluated;
        # density 100-400 kg/m3
        sd, rsn = x.new_like(), x.new_like()
        u = (torch.arange(n_src, device=dev, dtype=torch.float64) / n_src).unsqueeze(1).expand(n_src, L)
        sd.data[:, :L] = torch.where(u < 0.55, torch.zeros_like(u), torch.where(u < 0.9, 0.05 + u, 1e-4 * u)).to(tdt)
        rsn.data[:, :L] = (100.0 + 300.0 * torch.rand(n_src, L, device=dev)).to(tdt)
        del u
        record(f"combine snow_cover (2->1) {tag}", timeit(lambda: native.combine_stack(native.COMB_SNOW_COVER, [sd.data, rsn.data], [z.data],
               n_pts=n_src, n_lev=L, pitch=x.pitch, layout=COLUMNS)), 3 * stack_bytes, "regions: 55 % bare, 35 % deep snow, 10 % thin cover (tanh evaluated)")
        sd.data.fill_(1e-5)  # 4000 * (1000 * 1e-5 / rsn) / rsn < 2.65 everywhere: tanh on every element
        record(f"combine snow_cover (2->1) {tag}, thin cover everywhere", timeit(lambda: native.combine_stack(native.COMB_SNOW_COVER, [sd.data, rsn.data], [z.data],
               n_pts=n_src, n_lev=L, pitch=x.pitch, layout=COLUMNS)), 3 * stack_bytes, "worst case: tanh evaluated on every element")
        del sd, rsn
        record(f"combine difference (2->1, accum_to_interval) {tag}", timeit(lambda: native.combine_stack(native.COMB_SUB, [x.data, y.data], [z.data],
               n_pts=n_src, n_lev=L, pitch=x.pitch, layout=COLUMNS)), 3 * stack_bytes)
        record(f"combine cos_sin (1->2) {tag}", timeit(lambda: native.combine_stack(native.COMB_COS_SIN, [x.data], [y.data, z.data],
               n_pts=n_src, n_lev=L, pitch=x.pitch, layout=COLUMNS)), 3 * stack_bytes)
        # ---- the numpy-only domain filters (filters/domain.py)
        lim = torch.full((L,), 10000.0, dtype=torch.float64, device=dev)
        w2 = x.new_like()
        record(f"combine opera_clipping (2->2) {tag}", timeit(lambda: native.combine_stack(native.COMB_OPERA_CLIP, [x.data, y.data], [z.data, w2.data],
               n_pts=n_src, n_lev=L, pitch=x.pitch, layout=COLUMNS, level_param=lim)), 4 * stack_bytes)
        dm = x.new_like()
        dm.data[:, :L] = torch.randint(0, 4, (n_src, L), device=dev).to(tdt)
        record(f"combine opera_preprocessing (3->2) {tag}", timeit(lambda: native.combine_stack(native.COMB_OPERA_PREPROCESS, [x.data, y.data, dm.data], [z.data, w2.data],
               n_pts=n_src, n_lev=L, pitch=x.pitch, layout=COLUMNS, level_param=lim)), 5 * stack_bytes)
        del dm, w2
        # humidity conversions: q in [1e-6, 2e-2], t in [190, 320] K — water, mixed and ice branches of the saturation curve all present
        qh, th = x.new_like(), x.new_like()
        qh.data[:, :L] = (10.0 ** (-6.0 + 4.3 * torch.rand(n_src, L, device=dev))).to(tdt)
        th.data[:, :L] = (190.0 + 130.0 * torch.rand(n_src, L, device=dev)).to(tdt)
        plev = torch.linspace(1.0, 1000.0, L, dtype=torch.float64, device=dev)
        record(f"combine q_to_r (2->1) {tag}", timeit(lambda: native.combine_stack(native.COMB_Q_TO_R, [qh.data, th.data], [z.data],
               n_pts=n_src, n_lev=L, pitch=x.pitch, layout=COLUMNS, level_param=plev)), 3 * stack_bytes, "one or two exp per element (mixed phase)")
        record(f"combine r_to_d (2->1) {tag}", timeit(lambda: native.combine_stack(native.COMB_R_TO_D, [y.data, th.data], [z.data],
               n_pts=n_src, n_lev=L, pitch=x.pitch, layout=COLUMNS)), 3 * stack_bytes, "one exp and one log per element")
        del qh, th
        # one ORAS6 group: 14 fields and the ice concentration they share
        o_in, o_out = Stack.empty(n_src, 14, tdt, dev, COLUMNS), Stack.empty(n_src, 14, tdt, dev, COLUMNS)
        o_in.data.normal_()
        ice = torch.rand(n_src, device=dev).to(tdt)
        kinds = torch.tensor([0, 1, 1, 1, 4, 4, 1, 2, 3, 1, 1, 1, 2, 5], dtype=torch.float64, device=dev)
        record(f"combine oras6_clipping (14 fields of one date) {tag}", timeit(lambda: native.combine_stack(native.COMB_ORAS6, [o_in.data, ice], [o_out.data],
               n_pts=n_src, n_lev=14, pitch=o_in.pitch, layout=COLUMNS, level_param=kinds)), (2 * 14 + 1) * n_src * B,
               "one launch per group; the shared ice field is read once per point")
        cls = Stack.empty(n_src, 1, tdt, dev, COLUMNS)
        cls.data[:, 0] = torch.randint(0, 21, (n_src,), device=dev).to(tdt)
        val = cls.new_like()
        tab = torch.tensor([21.0] + [float(i) for i in range(21)], dtype=torch.float64, device=dev)
        record(f"combine lookup (land_parameters, 1 field) {tag}", timeit(lambda: native.combine_stack(native.COMB_LOOKUP, [cls.data], [val.data],
               n_pts=n_src, n_lev=1, pitch=cls.pitch, layout=COLUMNS, level_param=tab)), 2 * n_src * B, "26 MB / 53 MB: launch-latency sized")
        del o_in, o_out, ice, cls, val
        # ---- layout
        f = Stack.empty(n_src, L, tdt, dev, FIELDS)
        record(f"relayout columns->fields {tag}", timeit(lambda: native.relayout(x.data, f.data, n_pts=n_src, n_lev=L, src_pitch=x.pitch,
               dst_pitch=f.pitch, src_layout=COLUMNS, dst_layout=FIELDS)), 2 * stack_bytes)
        record(f"relayout fields->columns {tag}", timeit(lambda: native.relayout(f.data, y.data, n_pts=n_src, n_lev=L, src_pitch=f.pitch,
               dst_pitch=y.pitch, src_layout=FIELDS, dst_layout=COLUMNS)), 2 * stack_bytes)
        # ---- level gather (re-listing / sub-selecting the fields of a stack, operand stacks of the multi-input filters)
        half = Stack.empty(n_src, 68, tdt, dev, COLUMNS)
        record(f"select 68 of {L} levels (every other one) {tag}", timeit(lambda: native.select_levels(x.data, half.data, list(range(0, L - 1, 2)), n_pts=n_src, n_src_lev=L,
               src_pitch=x.pitch, dst_pitch=half.pitch, layout=COLUMNS)), 2 * n_src * 68 * B, "atx_select_levels; the source lines are read whole: 3x the algorithmic bytes")
        one_lev = Stack.empty(n_src, 1, tdt, dev, COLUMNS)
        record(f"select 1 of {L} levels {tag}", timeit(lambda: native.select_levels(x.data, one_lev.data, [77], n_pts=n_src, n_src_lev=L, src_pitch=x.pitch,
               dst_pitch=one_lev.pitch, layout=COLUMNS)), 2 * n_src * B, "one field out of a column stack: a 64-byte sector per point is the least that can move")
        del half, one_lev
        # ---- regrid on field-major
        pf = native.level_program([[(native.OP_AFFINE, 0, 1.0 + 0.001 * l, -273.15) for l in range(L)]], dev)
        g = f.new_like()
        kwf = dict(n_pts=n_src, n_lev=L, x_pitch=f.pitch, y_pitch=g.pitch, layout=FIELDS)
        record(f"pointwise affine, a scale per field {tag} fields (the reference's array order)",
               timeit(lambda: native.pointwise_stack(f.data, g.data, prog=pf, n_stage=1, **kwf)), 2 * stack_bytes, "one vector per lane, grid.y = field")
        del g
        record(f"regrid_ell k=4 {tag} fields", timeit(lambda: plan4.apply(f)), bench.algorithmic_bytes(L, B, U4, n_tgt, 4))
        # ---- masks / reductions on one field
        first = f.data[0].contiguous()
        mask = torch.empty(n_src + 8, dtype=torch.uint8, device=dev)
        record(f"mask_build {tag} (1 field)", timeit(lambda: native.mask_build(first, mask, n=n_src, cmp=native.CMP_GT, threshold=280.0)), n_src * (B + 1))
        record(f"reduce min {tag} (1 field)", timeit(lambda: native.reduce(first, native.RED_MIN)), n_src * B, "includes the device->host read of the result")
        del x, y, z, f
        torch.cuda.empty_cache()

    m = (torch.rand(n_src, device=dev) < 0.7).to(torch.uint8)
    cnt = int(m.sum().item())
    record("mask_to_index (6.6M points, 70% kept)", timeit(lambda: native.mask_to_index(m, n_src)), 2 * n_src + 4 * cnt, "includes workspace allocation + count read-back")

    # ---- k-NN precompute
    sxyz = torch.from_numpy(np.ascontiguousarray(interp.unit_sphere_xyz(src_grid["latitudes"], src_grid["longitudes"]))).to(dev)
    txyz = torch.from_numpy(np.ascontiguousarray(interp.unit_sphere_xyz(tgt_grid["latitudes"], tgt_grid["longitudes"]))).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index = native.KnnIndex(sxyz)
    torch.cuda.synchronize()
    res["knn_build O1280"] = {"ms": (time.perf_counter() - t0) * 1e3}
    for k in (1, 4):
        res[f"knn_query O1280->0.25 k={k}"] = {"ms": timeit(lambda: index.query(txyz, k), n=3, warm=1)}
    print({k: v for k, v in res.items() if k.startswith("knn")})
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
