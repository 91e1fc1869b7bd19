#!/usr/bin/env python3
"""Runs tools/experiments/gather_bounds.hip on the headline tables (O1280 -> 0.25 degree, k = 4, 137 levels, f32)."""
import ctypes, os, subprocess, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft, bench
graft.load_package()
from anemoi_transform_amd import interp
from anemoi_transform_amd.grids import lookup
from anemoi_transform_amd.stack import COLUMNS, Stack
so = os.path.join(HERE, "gather_bounds.so")
if not os.path.exists(so):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so,
                    os.path.join(HERE, "gather_bounds.hip")], check=True)
lib = ctypes.CDLL(so)
lib.run_gather.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
src_g, tgt_g = lookup("o1280"), lookup("0.25")
n_src, n_tgt, L = len(src_g["latitudes"]), len(tgt_g["latitudes"]), 137
idx64, w64 = interp.knn_inverse_distance(src_g, tgt_g, k=4, device=True)
idx = torch.from_numpy(idx64.astype(np.int32)).to(dev); w = torch.from_numpy(w64.astype(np.float32)).to(dev)
x = bench.synth_stack(src_g, L, torch.float32, dev, 0, COLUMNS); out = Stack.empty(n_tgt, L, torch.float32, dev, COLUMNS)
st = torch.cuda.current_stream().cuda_stream
U = int(np.unique(idx64).size)
read_gb = (L * 4 * U + n_tgt * 4 * 8) / 1e9; write_gb = L * 4 * n_tgt / 1e9
for mode, name, gb in ((0, "gather + store (the kernel)", read_gb + write_gb), (1, "gather only", read_gb), (2, "store only", write_gb)):
    ms, _ = bench.time_launches(lambda: lib.run_gather(mode, x.data.data_ptr(), out.data.data_ptr(), idx.data_ptr(), w.data_ptr(), n_tgt, 35, x.pitch, out.pitch, st), 20, 3)
    print(f"{name:30s} {ms:.4f} ms   {gb:.3f} GB algorithmic -> {gb / ms:.2f} TB/s", flush=True)
