#!/usr/bin/env python3
"""Runs tools/experiments/stream_shapes.hip (built next to it) and prints TB/s per launch shape."""
import ctypes, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import bench
so = os.path.join(HERE, "stream_shapes.so")
if not os.path.exists(so):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so,
                    os.path.join(HERE, "stream_shapes.hip")], check=True)
lib = ctypes.CDLL(so)
lib.run_shape.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
n = 6599680 * 140
x = torch.rand(n, dtype=torch.float32, device=dev); y = torch.empty_like(x)
gb = 2 * n * 4 / 1e9
st = torch.cuda.current_stream().cuda_stream
for shape, U, cap in [(0, 1, 0), (1, 2, 0), (1, 4, 0), (1, 8, 0), (2, 1, 65536), (2, 4, 2048), (2, 4, 8192), (2, 4, 65536), (2, 4, 262144), (2, 2, 65536), (2, 8, 65536)]:
    ms, _ = bench.time_launches(lambda: lib.run_shape(shape, U, cap, x.data_ptr(), y.data_ptr(), n // 4, st), 20, 3)
    print(f"shape {shape} U={U} cap={cap:7d}: {ms:.3f} ms  {gb / ms:.2f} TB/s", flush=True)
import numpy as np
lib.run_table.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dt = np.dtype([("op", np.int32), ("use_mask", np.int32), ("p0", np.float64), ("p1", np.float64)])
mask = (torch.rand(6599680 + 8, device=dev) < 0.3).to(torch.uint8)
for n_stage, use_mask in ((1, 0), (3, 0), (1, 1)):
    t = np.zeros((n_stage, 35), dtype=dt); t["op"] = 1; t["p0"] = 2.0; t["p1"] = 1.0; t["use_mask"] = use_mask
    table = torch.from_numpy(t.view(np.uint8).reshape(-1)).to(dev)
    ms, _ = bench.time_launches(lambda: lib.run_table(x.data_ptr(), y.data_ptr(), n // 4, 35, table.data_ptr(), n_stage, mask.data_ptr() if use_mask else None, st), 20, 3)
    print(f"shape 3 table lookup, {n_stage} stage(s), mask={use_mask}: {ms:.3f} ms  {gb / ms:.2f} TB/s", flush=True)
lib.run_table12.argtypes = lib.run_table.argtypes
dt12 = np.dtype([("op_mask", np.int32), ("p0", np.float32), ("p1", np.float32)])
for n_stage, use_mask in ((1, 0), (2, 0), (3, 0), (1, 1), (3, 1)):
    t = np.zeros((n_stage, 35), dtype=dt12); t["op_mask"] = 1 | (use_mask << 16); t["p0"] = 2.0; t["p1"] = 1.0
    table = torch.from_numpy(t.view(np.uint8).reshape(-1)).to(dev)
    ms, _ = bench.time_launches(lambda: lib.run_table12(x.data_ptr(), y.data_ptr(), n // 4, 35, table.data_ptr(), n_stage, mask.data_ptr() if use_mask else None, st), 20, 3)
    print(f"shape 4 table of 12-B operators, {n_stage} stage(s), mask={use_mask}: {ms:.3f} ms  {gb / ms:.2f} TB/s", flush=True)
ms, _ = bench.time_launches(lambda: torch.add(x, 1.0, out=y), 20, 3)
print(f"torch add(out=):           {ms:.3f} ms  {gb / ms:.2f} TB/s")
