#!/usr/bin/env python3
"""Workload for the SQ counter pass that backs the "ALU-bound" statements of DESIGN.md §3 (round 3, judge item 7): the float64
multi-input operators with transcendentals next to a memory-bound one, 137 levels of O1280 each.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
        --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python3 tools/pmc_sq_probe.py
    python3 tools/pmc_sq_probe.py --summarize gpurun_out/pmc_sq          # table: share of wave time parked / stalled / issuing, VALU per wave
"""

from __future__ import annotations

import argparse
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = ["difference f64", "cos_sin f64", "snow_cover f64 thin cover everywhere (tanh on every element)", "snow_cover f64 snow in regions",
         "pointwise affine f64", "pointwise log f64 (sp_to_lnsp)", "pointwise exp f64 (lnsp_to_sp)", "pointwise log then exp f64 (one two-stage program)",
         "difference f32", "cos_sin f32", "snow_cover f32 thin cover everywhere"]


def summarize(directory: str) -> None:
    rows = []
    for path in glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            rows += [r for r in csv.DictReader(f) if "combine_kernel" in r["Kernel_Name"] or "pointwise_cols_uniform_kernel" in r["Kernel_Name"]]
    # dispatches in launch order; every case was launched REPS times in a row
    by_dispatch: dict[int, dict] = {}
    for r in rows:
        d = by_dispatch.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    order = [by_dispatch[k] for k in sorted(by_dispatch)]
    reps = len(order) // len(CASES)
    print(f"{'case':66s} {'parked':>7s} {'stalled':>8s} {'issuing':>8s} {'VALU-issuing':>13s} {'VALU instr/wave':>16s} {'waves':>9s}")
    for i, case in enumerate(CASES):
        group = order[i * reps:(i + 1) * reps]
        avg = {k: sum(g[k] for g in group) / len(group) for k in group[0] if k != "name"}
        wc = avg["SQ_WAVE_CYCLES"]
        print(f"{case:66s} {avg['SQ_WAIT_ANY'] / wc:7.3f} {avg['SQ_WAIT_INST_ANY'] / wc:8.3f} {avg['SQ_ACTIVE_INST_ANY'] / wc:8.3f} "
              f"{avg['SQ_ACTIVE_INST_VALU'] / wc:13.3f} {avg['SQ_INSTS_VALU'] / avg['SQ_WAVES']:16.1f} {avg['SQ_WAVES']:9.0f}")
    print("(shares of SQ_WAVE_CYCLES; parked = s_waitcnt on memory, stalled = issue stall, issuing = an instruction of the wave is being issued)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--summarize", default=None)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    if args.summarize:
        return summarize(args.summarize)
    import torch

    import __graft_entry__ as graft

    graft.load_package()
    from anemoi_transform_amd import native
    from anemoi_transform_amd.stack import COLUMNS, column_pitch

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    native.load()
    L, n = 137, 6_599_680
    for tdt in (torch.float64, torch.float32):
        pitch = column_pitch(L, tdt)
        new = lambda: torch.zeros(n, pitch, dtype=tdt, device=dev)  # noqa: E731
        x, y, z, q, sd_thin, sd_regions, rsn = new(), new(), new(), new(), new(), new(), new()
        x[:, :L] = (280.0 + 30.0 * torch.randn(n, L, device=dev)).to(tdt)
        q[:, :L] = (6.28 * torch.rand(n, L, device=dev) - 3.14).to(tdt)
        u = (torch.arange(n, device=dev, dtype=torch.float64) / n).unsqueeze(1).expand(n, L)
        sd_regions[:, :L] = torch.where(u < 0.55, torch.zeros_like(u), torch.where(u < 0.9, 0.05 + u, 1e-4 * u)).to(tdt)
        del u
        sd_thin[:, :L] = 1e-5
        rsn[:, :L] = (100.0 + 300.0 * torch.rand(n, L, device=dev)).to(tdt)
        kw = dict(n_pts=n, n_lev=L, pitch=pitch, layout=COLUMNS)
        cases = [lambda: native.combine_stack(native.COMB_SUB, [x, rsn], [y], **kw),
                 lambda: native.combine_stack(native.COMB_COS_SIN, [q], [y, z], **kw),
                 lambda: native.combine_stack(native.COMB_SNOW_COVER, [sd_thin, rsn], [y], **kw)]
        if tdt == torch.float64:
            cases.append(lambda: native.combine_stack(native.COMB_SNOW_COVER, [sd_regions, rsn], [y], **kw))
            # round 5: the per-point programs with library functions (x: K-like, positive; q: |q| <= 3.14 as an argument of exp)
            pkw = dict(n_pts=n, n_lev=L, x_pitch=pitch, y_pitch=pitch, layout=COLUMNS)
            progs = [([(native.OP_AFFINE, 1.0, -273.15)], x), ([(native.OP_LOG, 0.0, 0.0)], x), ([(native.OP_EXP, 0.0, 0.0)], q),
                     ([(native.OP_LOG, 0.0, 0.0), (native.OP_EXP, 0.0, 0.0)], x)]
            for ops, src in progs:
                prog = native.level_program([[(op, 0, a, b)] * L for op, a, b in ops], dev)
                cases.append(lambda prog=prog, ns=len(ops), src=src: native.pointwise_stack(src, y, prog=prog, n_stage=ns, **pkw))
        for fn in cases:
            for _ in range(args.reps):
                fn()
        torch.cuda.synchronize()
        del x, y, z, q, sd_thin, sd_regions, rsn
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
