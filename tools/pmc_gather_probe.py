#!/usr/bin/env python3
"""Memory-side counters of the gather kernels (why long rows sit at 0.55-0.59 while their HBM traffic is close to algorithmic):

    for SET in "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_GATE_EN1" \\
               "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS" \\
               "TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY TCP_TCR_TCP_STALL_CYCLES" \\
               "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES"; do
        rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/pmc_gather/$i -- python3 tools/pmc_gather_probe.py
    python3 tools/pmc_gather_probe.py --summarize gpurun_out/pmc_gather

Cases (float32, O1280 -> 0.25 deg, 137 levels, REPS launches each, in this order): k = 4; k = 16 natural order; k = 16 in column blocks."""
from __future__ import annotations

import argparse
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = ["k=4 natural", "k=16 natural", "k=16 column blocks"]


def summarize(directory: str) -> None:
    per_case: dict[str, dict[str, float]] = {c: {} for c in CASES}
    n_xcd, n_cu = 8, 256
    for path in glob.glob(os.path.join(directory, "**", "*_agent_info.csv"), recursive=True):
        for a in csv.DictReader(open(path, newline="")):
            if a.get("Agent_Type") == "GPU":
                n_xcd, n_cu = int(a["Num_Xcc"]), int(a["Cu_Count"])
    for path in glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(path, newline="")) if "regrid_cols_ell_direct_kernel" in r["Kernel_Name"]]
        by_dispatch: dict[int, dict[str, float]] = {}
        for r in rows:
            by_dispatch.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        order = [by_dispatch[k] for k in sorted(by_dispatch)][len(CASES):]  # (the first launch of every plan uploads its tables: not counted)
        reps = len(order) // len(CASES)
        for i, case in enumerate(CASES):
            group = order[i * reps:(i + 1) * reps]
            for key in group[0]:
                per_case[case][key] = sum(g[key] for g in group) / len(group)
                if "GRBM_GUI_ACTIVE" in group[0] and key != "GRBM_GUI_ACTIVE":  # normalised by the GPU-active cycles of ITS OWN pass (run-to-run drift)
                    per_case[case][key + " / active"] = sum(g[key] / (g["GRBM_GUI_ACTIVE"] / n_xcd) for g in group) / len(group)
    keys = sorted({k for c in per_case.values() for k in c})
    print(f"{'counter':34s}" + "".join(f"{c:>22s}" for c in CASES))
    for k in keys:
        print(f"{k:34s}" + "".join(f"{per_case[c].get(k, float('nan')):22.4g}" for c in CASES))
    for c in CASES:
        v = per_case[c]
        line = [c + ":"]
        if "TCP_TCC_READ_REQ" in v and "TCP_TOTAL_CACHE_ACCESSES" in v:
            line.append(f"L1 read requests to L2 per L1 access {v['TCP_TCC_READ_REQ'] / v['TCP_TOTAL_CACHE_ACCESSES']:.3f}")
        if "TCP_UTCL1_TRANSLATION_MISS" in v and "TCP_UTCL1_REQUEST" in v:
            line.append(f"UTCL1 translation misses per request {v['TCP_UTCL1_TRANSLATION_MISS'] / v['TCP_UTCL1_REQUEST']:.4f}")
        if "TCP_TCC_READ_REQ_LATENCY" in v and "TCP_TCC_READ_REQ" in per_case[c]:
            line.append(f"mean L1->L2 read latency {v['TCP_TCC_READ_REQ_LATENCY'] / per_case[c]['TCP_TCC_READ_REQ']:.0f} cycles")
        if "TCC_HIT" in v and "TCC_MISS" in v:
            line.append(f"L2 hit rate {v['TCC_HIT'] / (v['TCC_HIT'] + v['TCC_MISS']):.3f}")
        if "TA_TA_BUSY" in v and "GRBM_GUI_ACTIVE" in v:
            # raw counters arrive summed over their instances: GRBM_GUI_ACTIVE over the XCDs, TA_* / TD_* over the TAs / TDs (one per active CU)
            active = v["GRBM_GUI_ACTIVE"] / n_xcd
            busy = v.get("TA_TA_BUSY / active", v["TA_TA_BUSY"] / active) / n_cu
            line.append(f"GPU-active cycles per XCD {active:.4g}; TA busy / GPU active, mean over the {n_cu} TAs {busy:.3f}")
            if "TA_BUSY_max / active" in v:
                line.append(f"busiest TA {v['TA_BUSY_max / active']:.3f}, idlest {v.get('TA_BUSY_min / active', float('nan')):.3f}")
            if "TA_ADDR_STALLED_BY_TC_CYCLES / active" in v:
                line.append(f"TA address stalled by TC {v['TA_ADDR_STALLED_BY_TC_CYCLES / active'] / n_cu:.3f} of GPU-active cycles, "
                            f"data stalled by TC {v['TA_DATA_STALLED_BY_TC_CYCLES / active'] / n_cu:.3f}")
            if "TA_ADDR_STALLED_BY_TD_CYCLES / active" in v:
                line.append(f"address stalled by TD {v['TA_ADDR_STALLED_BY_TD_CYCLES / active'] / n_cu:.3f}")
            if "TA_FLAT_READ_WAVEFRONTS / active" in v:
                line.append(f"GPU-active cycles per read wavefront-instruction and TA {n_cu / v['TA_FLAT_READ_WAVEFRONTS / active']:.1f}")
            if "TD_TD_BUSY / active" in v:
                line.append(f"TD busy / GPU active {v['TD_TD_BUSY / active'] / n_cu:.3f}, TD stalled by TC {v['TD_TC_STALL / active'] / n_cu:.3f} of GPU-active cycles")
        if "SQ_WAVE_CYCLES" in v:
            line.append(f"waves parked on memory {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}, issuing {v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f}")
        print("  ".join(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--summarize", default=None)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    if args.summarize:
        return summarize(args.summarize)
    import numpy as np
    import torch

    import __graft_entry__ as graft
    import bench

    graft.load_package()
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.gather import GatherPlan, target_order_for
    from anemoi_transform_amd.grids import lookup
    from anemoi_transform_amd.stack import COLUMNS

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    src, tgt = lookup("o1280"), lookup("0.25")
    n, nt = len(src["latitudes"]), len(tgt["latitudes"])
    idx16, w16 = interp.knn_inverse_distance(src, tgt, k=16, device=True, ties="index")
    idx4, w4 = idx16[:, :4], w16[:, :4] / w16[:, :4].sum(axis=1, keepdims=True)
    dtype = torch.float64 if os.environ.get("ATX_PROBE_DTYPE", "f32") == "f64" else torch.float32  # the same three cases in the headline's own width
    x = bench.synth_stack(src, 137, dtype, dev, 0, COLUMNS)
    plans = [GatherPlan(n, nt, index=idx4, weights=w4), GatherPlan(n, nt, index=idx16, weights=w16), GatherPlan(n, nt, index=idx16, weights=w16)]
    plans[2].order_targets(target_order_for(tgt["latitudes"], tgt["longitudes"], 16))
    for plan in plans:
        plan.apply(x)  # tables uploaded outside the counted launches?  (they are counted per dispatch of the regrid kernel only)
    torch.cuda.synchronize()
    for plan in plans:
        for _ in range(args.reps):
            out = plan.apply(x)
        torch.cuda.synchronize()
        del out


if __name__ == "__main__":
    main()
