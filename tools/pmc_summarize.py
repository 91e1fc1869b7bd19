#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes of tools/pmc_probe.py into profiles/traffic.json.

FETCH_SIZE / WRITE_SIZE are reported in KiB.  Following MI355X_MICROARCH.md §HBM the
read counter is calibrated on a launch with a known byte count in the same access
width (the full-stack copy of pmc_probe.py): correction = known_read_bytes /
(FETCH_SIZE * 1024); the guide's expectation for 16 B/lane streams on gfx950 is ~2.
The corrected per-launch HBM bytes of the regrid kernel are
    corrected_fetch = FETCH_SIZE * 1024 * read_correction
    corrected_write = WRITE_SIZE * 1024 * write_correction
"""

from __future__ import annotations

import argparse
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(directory: str, counter: str) -> dict[str, list[float]]:
    out: dict[str, list[float]] = {}
    for path in glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"]
                for key in ("stream_copy_kernel", "pointwise_cols_kernel", "pointwise_cols_flat_kernel", "pointwise_cols_table_kernel", "pointwise_fields_kernel",
                            "pointwise_cols_uniform_kernel", "regrid_cols_ell_direct_kernel", "regrid_cols_ell_kernel", "regrid_fields_ell_kernel",
                            "regrid_cols_csr_kernel", "pointwise_cols_levels_kernel", "pointwise_fields_rows_kernel"):
                    if key in name:
                        out.setdefault(key, []).append(float(row["Counter_Value"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch-dir", default=os.path.join(ROOT, "gpurun_out", "pmc_fetch"))
    ap.add_argument("--write-dir", default=os.path.join(ROOT, "gpurun_out", "pmc_write"))
    ap.add_argument("--meta", default=os.path.join(ROOT, "gpurun_out", "pmc_meta.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "traffic.json"))
    ap.add_argument("--round", default="r02")
    args = ap.parse_args()

    meta = json.load(open(args.meta))
    fetch = counters(args.fetch_dir, "FETCH_SIZE")
    write = counters(args.write_dir, "WRITE_SIZE")
    cal_f = fetch[meta["calibration_kernel"]][0] * 1024
    cal_w = write[meta["calibration_kernel"]][0] * 1024
    read_corr = meta["calibration_read_bytes"] / cal_f
    write_corr = meta["calibration_write_bytes"] / cal_w
    kern = meta["regrid_kernel"]
    # per repetition of the workload: a repetition may be several dispatches (level-chunk launches)
    reps = meta.get("regrid_launches") or len(fetch[kern])
    f_raw = sum(fetch[kern]) / reps * 1024
    w_raw = sum(write[kern]) / reps * 1024
    rec = {
        "round": args.round,
        "measured": f"round {args.round.lstrip('r0') or '0'}",
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/pmc_probe.py",
        "calibration": {
            "kernel": meta["calibration_kernel"],
            "known_read_bytes": meta["calibration_read_bytes"],
            "FETCH_SIZE_bytes": cal_f,
            "read_correction": read_corr,
            "known_write_bytes": meta["calibration_write_bytes"],
            "WRITE_SIZE_bytes": cal_w,
            "write_correction": write_corr,
        },
        "kernel": kern,
        "launches_averaged": reps,
        "dispatches_per_launch": len(fetch[kern]) / reps,
        "FETCH_SIZE_bytes_raw": f_raw,
        "WRITE_SIZE_bytes_raw": w_raw,
        "fetch_bytes_corrected": f_raw * read_corr,
        "write_bytes_corrected": w_raw * write_corr,
        "hbm_bytes_per_launch": f_raw * read_corr + w_raw * write_corr,
        "algorithmic_bytes_per_launch": meta["algorithmic_bytes_per_launch"],
    }
    rec["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
    cols = next((k for k in ("pointwise_cols_uniform_kernel", "pointwise_cols_table_kernel", "pointwise_cols_flat_kernel", "pointwise_cols_kernel") if k in fetch), None)
    if cols and "cols_copy_bytes" in meta:  # the column-stack per-point copy, same correction
        rec["pointwise_cols_copy"] = {
            "kernel": cols,
            "known_bytes_each_way": meta["cols_copy_bytes"],
            "fetch_over_known": fetch[cols][0] * 1024 * read_corr / meta["cols_copy_bytes"],
            "write_over_known": write[cols][0] * 1024 * write_corr / meta["cols_copy_bytes"],
        }
    table = {}
    if os.path.exists(args.out):
        table = json.load(open(args.out))
    table[meta["config"]] = rec
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(table, open(args.out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
