"""BASELINE.json configs 2, 4 and 5 as parity cases on one MI355X (config 3 is tests/test_gpu_fullsize.py
and the bench; config 1 is tests/test_filters.py::test_config1_32x64_single_field_plumbing).

  config 2  O96 -> 1 degree lat-lon, bilinear (k = 4 matrix in the reference's npz format), 1 surface field
  config 4  O1280 -> N320-sized target (542 080 points; the classic N320 row table is a downloaded data file, so the rows
            are constructed locally — grids.sized_row_lengths), 6 variables x 137 levels x 4 timesteps = 24 stacks batched,
            target points sharded 8 ways (the shards run one after the other on this GPU).
  config 5  regrid + orog_to_z + unit convert chained on O2560-shaped fields (26.3 M points x 137 levels, 14.4 GB)
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

from anemoi_transform_amd import interp, native
from anemoi_transform_amd.gather import GatherPlan
from anemoi_transform_amd.grids import lookup
from anemoi_transform_amd.stack import COLUMNS, Stack
from oracle import oracle

pytestmark = pytest.mark.gpu


def synth(grid, n_lev, dev, seed, dtype=torch.float32):
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    n = len(grid["latitudes"])
    lat = torch.from_numpy(np.deg2rad(grid["latitudes"])).to(dev)
    lon = torch.from_numpy(np.deg2rad(grid["longitudes"])).to(dev)
    st = Stack.empty(n, n_lev, dtype, dev, COLUMNS, zero=True)
    base = 280.0 + 30.0 * torch.sin(lat) * torch.cos(2.0 * lon)
    for l in range(n_lev):
        st.data[:, l] = (base + 0.1 * l + torch.randn(n, dtype=torch.float64, device=dev, generator=gen)).to(dtype)
    return st


def test_config2_o96_to_1deg_bilinear(dev, tmp_path):
    from anemoi_transform_amd.fields import fieldlist_from_dicts
    from anemoi_transform_amd.filters import create_filter_by_name

    src, tgt = lookup("o96"), lookup([1.0, 1.0])
    matrix = interp.bilinear_octahedral(96, tgt)
    assert tuple(matrix["matrix_shape"]) == (65160, 40320) and len(matrix["matrix_data"]) == 260640
    path = str(tmp_path / "o96-to-1deg-bilinear.npz")
    interp.save_matrix_npz(path, matrix, src, tgt)
    rng = np.random.default_rng(20260630)
    lat, lon = np.deg2rad(src["latitudes"]), np.deg2rad(src["longitudes"])
    for np_dtype in (np.float64, np.float32):
        values = (280 + 30 * np.sin(lat) * np.cos(2 * lon) + rng.standard_normal(len(lat))).astype(np_dtype)
        fields = fieldlist_from_dicts([{"param": "2t", "values": values, "latitudes": src["latitudes"], "longitudes": src["longitudes"]}])
        out = create_filter_by_name("regrid", matrix=path).forward(fields)[0]
        want = oracle.csr_apply(matrix["matrix_data"].astype(np_dtype), matrix["matrix_indices"], matrix["matrix_indptr"],
                                matrix["matrix_shape"], values)
        got = out.to_numpy()
        assert got.dtype == np_dtype and got.shape == (65160,)
        if np_dtype == np.float64:
            assert np.array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-6)


def test_config4_batched_stacks_target_sharded_8_ways(dev):
    """BASELINE configs[3] as specified: O1280 -> N320-sized target (542 080 points, grids.sized_row_lengths), 6 variables x
    4 timesteps = 24 stacks of 137 levels (3 288 fields, 88.7 GB f32 resident), k = 4, target points cut into 8
    traffic-balanced shards.  Replaces the per-field loop R: filters/fields/regrid.py:204-208.  Checked: sample levels of
    several stacks against the oracle's ``csr_array @ x`` (bit-exact), and every stack's 8 shard outputs concatenated against
    the unsharded result (bit-exact)."""
    src, tgt = lookup("o1280"), lookup("n320-sized")
    n_src, n_tgt, n_lev, n_stack = len(src["latitudes"]), len(tgt["latitudes"]), 137, 24
    assert (n_src, n_tgt, n_stack * n_lev) == (6599680, 542080, 3288)
    idx, w = interp.knn_inverse_distance(src, tgt, k=4)  # cKDTree: the reference's own builder
    plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
    bounds = plan.bounds(8)
    shards = [plan.shard(r, 8) for r in range(8)]
    assert bounds[0] == 0 and bounds[-1] == n_tgt and sum(s.n_tgt for s in shards) == n_tgt
    assert all(s.n_tgt > 0.5 * n_tgt / 8 for s in shards)

    gen = torch.Generator(device=dev)
    stacks = []
    for stack_id in range(n_stack):  # variable v = stack_id // 4, timestep = stack_id % 4
        gen.manual_seed(4000 + stack_id)
        st = Stack.empty(n_src, n_lev, torch.float32, dev, COLUMNS, zero=True)
        st.data[:, :n_lev].normal_(250.0 + 5.0 * (stack_id // 4), 20.0, generator=gen)
        stacks.append(st)
    full = plan.apply_many(stacks)  # ONE batched launch per 16 stacks
    assert len(full) == n_stack and all(o.n_pts == n_tgt and o.n_lev == n_lev for o in full)

    indptr = np.arange(n_tgt + 1) * 4
    w32 = w.astype(np.float32).reshape(-1)
    for stack_id, level in ((0, 0), (0, 136), (7, 68), (16, 5), (23, 136)):
        want = oracle.csr_apply(w32, idx.reshape(-1), indptr, (n_tgt, n_src), stacks[stack_id].level_numpy(level))
        assert np.array_equal(full[stack_id].level_numpy(level), want), (stack_id, level)

    # the same fields with the 6 variables of one timestep sharing a column (ONE stack of 6 x 137 = 822 levels, what bench.py's
    # config-4 line times as its second layout): level for level the same bits — and an 822-level launch of the direct kernel
    timestep = 2
    members = [4 * v + timestep for v in range(6)]  # stack_id = 4 * variable + timestep
    tall = Stack.empty(n_src, 6 * n_lev, torch.float32, dev, COLUMNS, zero=True)
    for j, sid in enumerate(members):
        tall.data[:, j * n_lev:(j + 1) * n_lev] = stacks[sid].data[:, :n_lev]
    out_tall = plan.apply(tall)
    assert out_tall.n_lev == 822
    for j, sid in enumerate(members):
        assert torch.equal(out_tall.data[:, j * n_lev:(j + 1) * n_lev].contiguous().view(torch.int32),
                           full[sid].data[:, :n_lev].contiguous().view(torch.int32)), sid
    del tall, out_tall

    pieces = [s.apply_many(stacks) for s in shards]  # what ranks 0..7 would each compute
    for stack_id in range(n_stack):
        parts = torch.cat([pieces[r][stack_id].data for r in range(8)])
        assert parts.shape == full[stack_id].data.shape
        assert torch.equal(parts[:, :n_lev].contiguous().view(torch.int32), full[stack_id].data[:, :n_lev].contiguous().view(torch.int32))
    del pieces, full
    # a rank only needs the latitude band of the source its slice references: a contiguous slab of the column stack
    from anemoi_transform_amd.distributed import rebase_plan, source_band

    x = stacks[3]
    for r in (0, 3, 7):
        lo, hi = source_band(shards[r])
        assert (hi - lo) < 0.25 * n_src
        slab = Stack(x.data[lo:hi], hi - lo, x.n_lev, COLUMNS)
        assert torch.equal(rebase_plan(shards[r], lo, hi).apply(slab).data, shards[r].apply(x).data)


@pytest.mark.parametrize("tdtype,np_dtype", [(torch.float32, np.float32), (torch.float64, np.float64)], ids=["f32", "f64"])
def test_config5_chained_filters_on_o2560(dev, tdtype, np_dtype):
    """BASELINE configs[4]: regrid + orog_to_z + unit convert on an ERA5-shape O2560 stack, fused into one launch — in float32
    (14.4 GB) and in float64, the reference's own width (28.8 GB resident)."""
    src, tgt = lookup("o2560"), lookup("0.25")
    n_src, n_tgt, n_lev = len(src["latitudes"]), len(tgt["latitudes"]), 137
    assert n_src == 26306560
    idx, w = interp.knn_inverse_distance(src, tgt, k=4, device=True)  # the table cKDTree would build (tools/knn_table_parity.py shows it row by row); cKDTree alone needs ~1 min here
    plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
    x = synth(src, n_lev, dev, 5, dtype=tdtype)
    # levels 0..135: temperature -> degC; level 136: orography -> geopotential -> (no convert)
    s_orog = [(native.OP_COPY, 0, 0.0, 0.0)] * (n_lev - 1) + [(native.OP_MUL, 0, oracle.G, 0.0)]
    s_conv = [(native.OP_AFFINE, 0, 1.0, -273.15)] * (n_lev - 1) + [(native.OP_COPY, 0, 0.0, 0.0)]
    prog = native.level_program([s_orog, s_conv], dev)
    fused = plan.apply(x, prog=prog, n_stage=2)
    plain = plan.apply(x)
    chained = plain.new_like()
    native.pointwise_stack(plain.data, chained.data, n_pts=n_tgt, n_lev=n_lev, x_pitch=plain.pitch, y_pitch=chained.pitch,
                           layout=COLUMNS, prog=prog, n_stage=2)
    bits = torch.int32 if tdtype == torch.float32 else torch.int64
    assert fused.data.dtype == tdtype
    assert torch.equal(fused.data[:, :n_lev].contiguous().view(bits), chained.data[:, :n_lev].contiguous().view(bits))
    indptr = np.arange(n_tgt + 1) * 4
    weights = w.astype(np_dtype).reshape(-1)
    for l, fn in ((0, lambda v: oracle.rescale_forward(v, np_dtype(1.0), np_dtype(-273.15))), (136, lambda v: v * np_dtype(oracle.G))):
        base = oracle.csr_apply(weights, idx.reshape(-1), indptr, (n_tgt, n_src), x.level_numpy(l))
        want = fn(base)
        assert want.dtype == np_dtype
        assert np.array_equal(fused.level_numpy(l), want)
    del x, fused, plain, chained
    torch.cuda.empty_cache()


def test_config5_three_variables_sharing_a_column(dev):
    """BASELINE configs[4] as SURVEY.md §8d words it — 137 levels x {t, orography-like, one more variable} — with the three variables of
    a grid point in ONE column (a 411-level O2560 stack, 43 GB float32): the fused launch (program with three runs of levels: the
    direct kernel's typed per-level table) against regrid-then-program and, on sample levels, against the oracle."""
    src, tgt = lookup("o2560"), lookup("0.25")
    n_src, n_tgt, n1 = len(src["latitudes"]), len(tgt["latitudes"]), 137
    n_lev = 3 * n1
    idx, w = interp.knn_inverse_distance(src, tgt, k=4, device=True)
    plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
    x = Stack.empty(n_src, n_lev, torch.float32, dev, COLUMNS, zero=True)
    gen = torch.Generator(device=dev)
    gen.manual_seed(55)
    x.data[:, :n_lev].normal_(270.0, 15.0, generator=gen)
    cp, to_z, to_degc = (native.OP_COPY, 0, 0.0, 0.0), (native.OP_MUL, 0, oracle.G, 0.0), (native.OP_AFFINE, 0, 1.0, -273.15)
    prog = native.level_program([[cp] * n1 + [to_z] * n1 + [cp] * n1, [to_degc] * n1 + [cp] * (2 * n1)], dev)
    fused = plan.apply(x, prog=prog, n_stage=2)
    chained = plan.apply(x)
    native.pointwise_stack(chained.data, chained.data, n_pts=n_tgt, n_lev=n_lev, x_pitch=chained.pitch, y_pitch=chained.pitch, layout=COLUMNS,
                           prog=prog, n_stage=2)
    assert torch.equal(fused.data[:, :n_lev].contiguous().view(torch.int32), chained.data[:, :n_lev].contiguous().view(torch.int32))
    indptr = np.arange(n_tgt + 1) * 4
    w32 = w.astype(np.float32).reshape(-1)
    for l, fn in ((5, lambda v: oracle.rescale_forward(v, np.float32(1.0), np.float32(-273.15))), (n1 + 70, lambda v: v * np.float32(oracle.G)), (n_lev - 1, lambda v: v)):
        base = oracle.csr_apply(w32, idx.reshape(-1), indptr, (n_tgt, n_src), x.level_numpy(l))
        assert np.array_equal(fused.level_numpy(l), fn(base)), l
    del x, fused, chained
    torch.cuda.empty_cache()


def test_kernels_on_a_stack_beyond_2_to_31_elements(dev):
    """O2560 x 137 levels = 3.6e9 elements (14.4 GB f32): every streaming kernel must do its index arithmetic in 64 bits.
    Checked on samples against torch indexing (bit copies) and against an independent reduction."""
    from anemoi_transform_amd.stack import FIELDS

    n_pts, n_lev = 26306560, 137
    assert n_pts * 140 > 2**31
    x = Stack.empty(n_pts, n_lev, torch.float32, dev, COLUMNS)
    x.data.copy_((torch.arange(n_pts, device=dev, dtype=torch.float32) % 8191.0).unsqueeze(1) * 0.25)  # exact in f32
    x.data[:, :n_lev] += torch.arange(n_lev, device=dev, dtype=torch.float32) * 2048.0
    probe = torch.tensor([0, 1, 12345, n_pts // 2, n_pts - 2, n_pts - 1], device=dev)
    # per-point program over the whole stack (flat kernel) and its in-place form
    y = x.new_like()
    prog = native.level_program([[(native.OP_AFFINE, 0, 2.0, 1.0)] * n_lev], dev)
    kw = dict(n_pts=n_pts, n_lev=n_lev, x_pitch=x.pitch, y_pitch=y.pitch, layout=COLUMNS, prog=prog, n_stage=1)
    native.pointwise_stack(x.data, y.data, **kw)
    assert torch.equal(y.data[probe, :n_lev], x.data[probe, :n_lev] * 2.0 + 1.0)
    assert torch.equal(y.data[-5:, :n_lev], x.data[-5:, :n_lev] * 2.0 + 1.0)
    # reduction over the pitched stack
    assert native.reduce_stack(x.data, native.RED_MAX, n_pts=n_pts, n_lev=n_lev, pitch=x.pitch, layout=COLUMNS) == 8190 * 0.25 + 136 * 2048.0
    assert native.reduce_stack(x.data, native.RED_NANCOUNT, n_pts=n_pts, n_lev=n_lev, pitch=x.pitch, layout=COLUMNS) == 0.0
    # level gather of the last levels, then layout conversion both ways
    sel = Stack.empty(n_pts, 3, torch.float32, dev, COLUMNS)
    native.select_levels(x.data, sel.data, [136, 0, 68], n_pts=n_pts, n_src_lev=n_lev, src_pitch=x.pitch, dst_pitch=sel.pitch, layout=COLUMNS)
    assert torch.equal(sel.data[probe, :3], x.data[probe][:, [136, 0, 68]])
    del y, sel
    f = x.to_layout(FIELDS)
    assert torch.equal(f.data[:, probe], x.data[probe, :n_lev].T) and torch.equal(f.data[136, -3:], x.data[-3:, 136])
    back = f.to_layout(COLUMNS)
    assert torch.equal(back.data[probe, :n_lev], x.data[probe, :n_lev]) and torch.equal(back.data[-3:, :n_lev], x.data[-3:, :n_lev])
    del f
    # multi-input kernel: difference of two stacks
    z = x.new_like()
    native.combine_stack(native.COMB_SUB, [back.data, x.data], [z.data], n_pts=n_pts, n_lev=n_lev, pitch=x.pitch, layout=COLUMNS)
    assert native.reduce_stack(z.data, native.RED_MAX, n_pts=n_pts, n_lev=n_lev, pitch=z.pitch, layout=COLUMNS) == 0.0
    assert native.reduce_stack(z.data, native.RED_MIN, n_pts=n_pts, n_lev=n_lev, pitch=z.pitch, layout=COLUMNS) == 0.0
