#!/usr/bin/env python3
"""Per-launch-shape averages of one kernel from a rocprofv3 --kernel-trace CSV.

`rocprofv3 --stats` averages every dispatch of a kernel NAME; bench.py's default run launches the headline kernel in several
shapes (the 422 headline launches, and the batched / sharded launches of extras.config4), so the figure that has to agree
with bench.py's HIP-event average is the average over the dispatches of the HEADLINE shape.  This prints both.

    python tools/kernel_trace_by_shape.py gpurun_out/prof_r02/*/*_kernel_trace.csv regrid_cols_ell_direct_kernel
"""

from __future__ import annotations

import collections
import csv
import sys


def main():
    path, needle = sys.argv[1], sys.argv[2]
    shapes = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if needle in row["Kernel_Name"]:
            key = (row["Kernel_Name"].split("(")[0].replace("void ", ""), int(row["Grid_Size_X"]), int(row["Grid_Size_Y"]), int(row["Workgroup_Size_X"]))
            shapes[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print("kernel,grid_x_lanes,grid_y,workgroup,calls,average_ns,min_ns,max_ns")
    for (name, gx, gy, wg), ns in sorted(shapes.items(), key=lambda kv: -len(kv[1])):
        print(f"\"{name}\",{gx},{gy},{wg},{len(ns)},{sum(ns) / len(ns):.1f},{min(ns)},{max(ns)}")


if __name__ == "__main__":
    main()
