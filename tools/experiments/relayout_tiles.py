#!/usr/bin/env python3
"""atx_relayout both ways over 137 levels of O1280 — run once per library build (tile shapes: ATX_TP_BYTES, ATX_TP_LC)."""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as graft  # noqa: E402
from per_level_programs import launches  # noqa: E402


def main():
    graft.load_package()
    from anemoi_transform_amd import native
    from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n, L = 6599680, 137
    out = []
    for tdt, B, tag in ((torch.float32, 4, "f32"), (torch.float64, 8, "f64")):
        c = Stack.empty(n, L, tdt, dev, COLUMNS)
        c.data.normal_()
        f = Stack.empty(n, L, tdt, dev, FIELDS)
        kw = dict(n_pts=n, n_lev=L)
        ms_cf = launches(lambda: native.relayout(c.data, f.data, src_pitch=c.pitch, dst_pitch=f.pitch, src_layout=COLUMNS, dst_layout=FIELDS, **kw))
        back = c.new_like()
        ms_fc = launches(lambda: native.relayout(f.data, back.data, src_pitch=f.pitch, dst_pitch=back.pitch, src_layout=FIELDS, dst_layout=COLUMNS, **kw))
        assert torch.equal(back.data[:, :L], c.data[:, :L])
        half = Stack.empty(n, 68, tdt, dev, COLUMNS)
        ms_sel = launches(lambda: native.select_levels(c.data, half.data, list(range(0, L - 1, 2)), n_pts=n, n_src_lev=L, src_pitch=c.pitch,
                                                       dst_pitch=half.pitch, layout=COLUMNS))
        assert torch.equal(half.data[:, :68], c.data[:, 0:136:2])
        out.append(f"{tag} c->f {ms_cf:.3f} ms {2 * n * L * B / (ms_cf * 1e-3) / 8e12:.3f} | f->c {ms_fc:.3f} ms {2 * n * L * B / (ms_fc * 1e-3) / 8e12:.3f} | select 68 of 137 {ms_sel:.3f} ms {2 * n * 68 * B / (ms_sel * 1e-3) / 8e12:.3f}")
        del c, f, back, half
        torch.cuda.empty_cache()
    print(os.path.basename(native.lib_path()), " || ".join(out), flush=True)


if __name__ == "__main__":
    main()
