import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as graft, bench
graft.load_package()
from anemoi_transform_amd import interp, native
from anemoi_transform_amd.gather import column_block_order
from anemoi_transform_amd.grids import lookup
from anemoi_transform_amd.stack import COLUMNS, Stack
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
src, tgt = lookup("o1280"), lookup("0.25")
n_src, n_tgt, L = len(src["latitudes"]), len(tgt["latitudes"]), 137
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
idx, w = interp.knn_inverse_distance(src, tgt, k=K, device=True, ties="index")
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for tdt, npdt in ((torch.float64, np.float64), (torch.float32, np.float32)):
    x = bench.synth_stack(src, L, tdt, dev, 0, COLUMNS)
    out = Stack.empty(n_tgt, L, tdt, dev, COLUMNS)
    for bp in (360, 256, 160):
        order = column_block_order(tgt["latitudes"], tgt["longitudes"], block_points=bp)
        ident = torch.arange(n_tgt, dtype=torch.int32, device=dev)
        cases = {
            "natural tables, no rows": (idx, w, None),
            "natural tables, identity rows": (idx, w, ident),
            "ordered tables, identity rows (sequential writes; wrong values)": (idx[order], w[order], ident),
            "ordered tables, true rows": (idx[order], w[order], torch.from_numpy(order).to(dev)),
        }
        for name, (i, ww, rows) in cases.items():
            i_d = torch.from_numpy(np.ascontiguousarray(i).astype(np.int32)).to(dev)
            w_d = torch.from_numpy(np.ascontiguousarray(ww).astype(npdt)).to(dev)
            ms = timeit(lambda: native.regrid_ell(x.data, out.data, i_d, w_d, n_src=n_src, n_tgt=n_tgt, k=K, n_lev=L, src_pitch=x.pitch, out_pitch=out.pitch, layout=COLUMNS, tgt_rows=rows))
            print(f"k={K:2d} {str(tdt)[6:]:8s} block_points {bp:3d}  {name:64s} {ms:.4f} ms", flush=True)
    del x, out
