"""Parity at BASELINE.json's full sizes (O1280 -> 0.25 degree, 137 levels) through
size-independent properties, plus oracle spot checks on a sample of levels.

The oracle cannot afford 137 full-size levels per test, so the full stack is checked by
properties the operator must have whatever its size:
  * rows of the index table that point at target t == source t reproduce the source (identity gather);
  * weights sum to 1  =>  a constant field stays constant;
  * linearity: R(a x + y) == a R(x) + R(y) up to rounding;
  * the 8 target shards concatenate to the unsharded result, bit for bit;
  * column-stack and field-major kernels agree bit for bit (same arithmetic order);
  * compaction indices are strictly increasing and select exactly the non-NaN points;
  * affine then inverse-affine returns the input to within 1 ulp-scale error;
and a sample of levels is compared with the oracle directly.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

from anemoi_transform_amd import native
from anemoi_transform_amd.gather import TARGET_COST, GatherPlan
from anemoi_transform_amd.grids import lookup
from anemoi_transform_amd.interp import knn_inverse_distance
from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack
from oracle import oracle

pytestmark = pytest.mark.gpu

N_LEV = 137


@pytest.fixture(scope="module")
def case(dev):
    src, tgt = lookup("o1280"), lookup("0.25")
    idx, w = knn_inverse_distance(src, tgt, k=4)
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    gen = torch.Generator(device=dev)
    gen.manual_seed(20260630)
    lat = torch.from_numpy(np.deg2rad(src["latitudes"])).to(dev)
    lon = torch.from_numpy(np.deg2rad(src["longitudes"])).to(dev)
    x = Stack.empty(n_src, N_LEV, torch.float32, dev, COLUMNS, zero=True)
    base = 280.0 + 30.0 * torch.sin(lat) * torch.cos(2.0 * lon)
    for l in range(N_LEV):
        x.data[:, l] = (base + 0.1 * l + torch.randn(n_src, dtype=torch.float64, device=dev, generator=gen)).float()
    return dict(src=src, tgt=tgt, idx=idx, w=w, n_src=n_src, n_tgt=n_tgt, x=x,
                plan=GatherPlan(n_src, n_tgt, index=idx, weights=w))


@pytest.fixture(scope="module")
def case64(case, dev):
    """The HEADLINE instantiation at the headline size: the float64 stack (the reference's own width, R: fields.py:178-202) that
    `bench.py` times — `regrid_cols_ell_direct_kernel<double, 2, 4, true, false, 0>` over 71.6 M (target, vector) items."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(20260631)
    lat = torch.from_numpy(np.deg2rad(case["src"]["latitudes"])).to(dev)
    lon = torch.from_numpy(np.deg2rad(case["src"]["longitudes"])).to(dev)
    x = Stack.empty(case["n_src"], N_LEV, torch.float64, dev, COLUMNS, zero=True)
    base = 280.0 + 30.0 * torch.sin(lat) * torch.cos(2.0 * lon)
    for l in range(N_LEV):
        x.data[:, l] = base + 0.1 * l + torch.randn(case["n_src"], dtype=torch.float64, device=dev, generator=gen)
    return dict(case, x=x)


def test_headline_float64_sample_levels_are_scipys_bits(case64):
    """R: regrid.py:310 `csr_array @ x` in float64 — BASELINE configs[2] in the width the bench line reports."""
    out = case64["plan"].apply(case64["x"])
    assert out.data.dtype == torch.float64 and out.n_pts == case64["n_tgt"] and out.n_lev == N_LEV
    indptr = np.arange(case64["n_tgt"] + 1) * 4
    for l in (0, 68, 136):
        want = oracle.csr_apply(case64["w"].reshape(-1), case64["idx"].reshape(-1), indptr, (case64["n_tgt"], case64["n_src"]),
                                case64["x"].level_numpy(l))
        assert want.dtype == np.float64
        assert np.array_equal(out.level_numpy(l), want), l  # bit-exact: scipy's summation order, no FMA


def test_headline_float64_through_the_plugin_api(case64, tmp_path):
    """BASELINE configs[2] by the DROP-IN route: `create_filter_by_name("regrid", matrix=<k = 4 npz>)` (R: filters/__init__.py:36-60,
    regrid.py:174-208 — the per-field loop — and :283-285,310 — `MIRMatrix`) on a FieldList of 137 device-resident float64 O1280
    fields.  Every output field must hold the bits `GatherPlan.apply` produces for its level (the route the other full-size tests and
    the bench take), three of them are compared with scipy directly, and the fields come back in list order with the target grid's
    coordinates and their own metadata."""
    from anemoi_transform_amd import interp
    from anemoi_transform_amd.fields import ArrayField, FieldList, new_field_from_stack
    from anemoi_transform_amd.filters import create_filter_by_name

    src, tgt, x = case64["src"], case64["tgt"], case64["x"]
    matrix = interp.ell_to_csr(case64["idx"], case64["w"], case64["n_src"])
    path = str(tmp_path / "o1280-to-0p25-knn4.npz")
    interp.save_matrix_npz(path, matrix, src, tgt)  # the reference's own file layout (R: regrid.py:281-290)
    template = ArrayField(np.zeros(1), {"param": "t"}, np.zeros(1), np.zeros(1))
    fields = FieldList([new_field_from_stack(x, l, template=template, latitudes=src["latitudes"], longitudes=src["longitudes"],
                                             metadata={"param": "t", "levelist": l + 1}) for l in range(N_LEV)])
    out = create_filter_by_name("regrid", matrix=path).forward(fields)
    assert len(out) == N_LEV
    direct = case64["plan"].apply(x)
    for l, f in enumerate(out):
        assert f.metadata("param") == "t" and f.metadata("levelist") == l + 1
        stack, level = f.stack_ref()  # still in HBM: no host round trip inside the filter
        assert stack.n_pts == case64["n_tgt"] and stack.data.dtype == torch.float64
        assert torch.equal(stack.data[:, level].view(torch.int64), direct.data[:, l].view(torch.int64)), l
    lat, lon = out[0].grid_points()
    assert np.array_equal(lat, tgt["latitudes"]) and np.array_equal(lon, tgt["longitudes"])
    indptr = np.arange(case64["n_tgt"] + 1) * 4
    for l in (0, 68, 136):
        want = oracle.csr_apply(case64["w"].reshape(-1), case64["idx"].reshape(-1), indptr, (case64["n_tgt"], case64["n_src"]), x.level_numpy(l))
        got = out[l].to_numpy()
        assert got.dtype == np.float64 and np.array_equal(got, want), l  # scipy's bits through Filter.forward()


def test_headline_meets_the_north_star_roofline_target(case64, dev):
    """BASELINE.json north_star: ">= 60 % of per-GPU HBM3E bandwidth on O1280 -> 0.25 degree 137-level regrid at 1 GPU" — the headline
    launch (float64, k = 4) and the k = 1 gather on the same stack, priced on SURVEY.md §8d's algorithmic bytes over the HIP-event duration
    of a single launch (the minimum of 20: see below; bench.py reports average, median and minimum) against 8 TB/s.  Measured 0.70-0.71 and 0.72 on every box of rounds 2-5; the
    floor asserted is the target itself."""
    n_src, n_tgt, x = case64["n_src"], case64["n_tgt"], case64["x"]
    out = Stack.empty(n_tgt, N_LEV, torch.float64, dev, COLUMNS)
    idx_d = torch.from_numpy(case64["idx"].astype(np.int32)).to(dev)
    w_d = torch.from_numpy(case64["w"]).to(dev)
    idx1_d = torch.from_numpy(np.ascontiguousarray(case64["idx"][:, 0]).astype(np.int32)).to(dev)

    def frac(fn, k, n_unique):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        # the FASTEST of the 20 launches: this is a functional suite on a box other processes may share — a co-tenant or a clock dip
        # lengthens some launches, it cannot shorten any, so the minimum is the kernel's own time (the bench line and its rocprof
        # record carry the averages: 0.694-0.720 on nine boxes of round 5)
        ms = float(np.min([a.elapsed_time(b) for a, b in evs]))
        alg = N_LEV * 8 * (n_unique + n_tgt) + n_tgt * k * 4 + (n_tgt * k * 8 if k > 1 else 0)
        return alg / (ms * 1e-3) / 8.0e12

    kw = dict(n_src=n_src, n_tgt=n_tgt, n_lev=N_LEV, src_pitch=x.pitch, out_pitch=out.pitch, layout=COLUMNS)
    k4 = frac(lambda: native.regrid_ell(x.data, out.data, idx_d, w_d, k=4, **kw), 4, int(np.unique(case64["idx"]).size))
    k1 = frac(lambda: native.regrid_ell(x.data, out.data, idx1_d, None, k=1, **kw), 1, int(np.unique(case64["idx"][:, 0]).size))
    assert k4 >= 0.60, f"headline at {k4:.3f} of the HBM peak"
    assert k1 >= 0.60, f"k = 1 gather at {k1:.3f} of the HBM peak"


def test_headline_float64_nearest_is_a_bit_copy(case64):
    """R: regrid.py:380 `x[..., nearest_grid_points]` at full size in float64 (the `extras.nearest_k1` launch of the bench)."""
    nearest = np.ascontiguousarray(case64["idx"][:, 0])
    plan = GatherPlan(case64["n_src"], case64["n_tgt"], index=nearest)
    out = plan.apply(case64["x"])
    for l in (0, 68, 136):
        want = oracle.gather_nn(case64["x"].level_numpy(l), nearest)
        assert np.array_equal(out.level_numpy(l).view(np.uint64), want.view(np.uint64)), l
    # and the whole stack against torch's own indexing, bit for bit
    picked = case64["x"].data[torch.from_numpy(nearest).to(out.data.device)]
    assert torch.equal(out.data[:, :N_LEV].contiguous().view(torch.int64), picked[:, :N_LEV].contiguous().view(torch.int64))


def test_headline_float64_shards_concatenate_bit_exact(case64):
    """The 8 traffic-balanced target shards of the float64 job (what ranks 0..7 of the 8-GPU run compute) against the one launch."""
    full = case64["plan"].apply(case64["x"]).data
    parts = [case64["plan"].shard(r, 8).apply(case64["x"]).data for r in range(8)]
    assert sum(p.shape[0] for p in parts) == case64["n_tgt"]
    assert torch.equal(torch.cat(parts)[:, :N_LEV].contiguous().view(torch.int64), full[:, :N_LEV].contiguous().view(torch.int64))


def test_sample_levels_match_oracle(case):
    out = case["plan"].apply(case["x"])
    indptr = np.arange(case["n_tgt"] + 1) * 4
    w32 = case["w"].astype(np.float32).reshape(-1)
    for l in (0, 68, 136):
        want = oracle.csr_apply(w32, case["idx"].reshape(-1), indptr, (case["n_tgt"], case["n_src"]), case["x"].level_numpy(l))
        got = out.level_numpy(l)
        np.testing.assert_allclose(got, want, rtol=1e-6)  # north_star tolerance for float interpolation
        assert np.array_equal(got, want)  # and in fact identical: same order, no FMA


def test_constant_field_is_preserved(case, dev):
    c = Stack.empty(case["n_src"], N_LEV, torch.float32, dev, COLUMNS, zero=True)
    levels = torch.arange(1, N_LEV + 1, dtype=torch.float32, device=dev) * 3.5
    c.data[:, :N_LEV] = levels
    out = case["plan"].apply(c)
    err = (out.data[:, :N_LEV] - levels).abs().max().item()
    assert err <= 4 * np.finfo(np.float32).eps * float(levels.max())  # sum of 4 f32-rounded weights


def test_linearity(case, dev):
    x = case["x"]
    y = Stack.empty(case["n_src"], N_LEV, torch.float32, dev, COLUMNS, zero=True)
    y.data[:, :N_LEV] = torch.rand(case["n_src"], N_LEV, device=dev) * 10.0
    a = 0.5  # exact in binary
    z = Stack(x.data * a + y.data, case["n_src"], N_LEV, COLUMNS)
    rz = case["plan"].apply(z).data[:, :N_LEV]
    combo = case["plan"].apply(x).data[:, :N_LEV] * a + case["plan"].apply(y).data[:, :N_LEV]
    assert torch.allclose(rz, combo, rtol=1e-6, atol=1e-4)


def test_identity_gather_is_a_bit_copy(case):
    n = case["n_tgt"]
    ident = GatherPlan(case["n_src"], n, index=np.arange(n))
    out = ident.apply(case["x"])
    assert torch.equal(out.data.view(torch.int32), case["x"].data[:n].view(torch.int32))
    rev = GatherPlan(case["n_src"], n, index=np.arange(n)[::-1].copy())
    out = rev.apply(case["x"])
    assert torch.equal(out.data.view(torch.int32), case["x"].data[:n].flip(0).view(torch.int32))


def test_shards_concatenate_bit_exact(case):
    full = case["plan"].apply(case["x"]).data
    parts = [case["plan"].shard(r, 8).apply(case["x"]).data for r in range(8)]
    assert torch.equal(torch.cat(parts).view(torch.int32), full.view(torch.int32))
    # shard boundaries are balanced by traffic: polar shards of the lat-lon target hold more targets
    sizes = [p.shape[0] for p in parts]
    assert sizes[0] > 1.3 * sizes[3] and sizes[7] > 1.3 * sizes[4] and sum(sizes) == case["n_tgt"]
    cost = [np.unique(case["idx"][b0:b1]).size + TARGET_COST * (b1 - b0) for b0, b1 in zip(case["plan"].bounds(8)[:-1], case["plan"].bounds(8)[1:])]
    assert max(cost) < 1.1 * min(cost)


def test_layouts_agree_bit_exact(case):
    cols = case["plan"].apply(case["x"])
    fields = case["plan"].apply(case["x"].to_layout(FIELDS))
    assert fields.layout == FIELDS
    assert torch.equal(fields.data[:, : case["n_tgt"]].T.contiguous().view(torch.int32),
                       cols.data[:, :N_LEV].contiguous().view(torch.int32))


def test_csr_path_equals_ell_path(case, dev):
    """The general CSR kernel on the same (uniform) matrix gives the ELL kernel's bits."""
    n_tgt = case["n_tgt"]
    csr = GatherPlan(case["n_src"], n_tgt, csr=(case["w"].reshape(-1), case["idx"].reshape(-1), np.arange(n_tgt + 1) * 4))
    a = csr.apply(case["x"]).data
    b = case["plan"].apply(case["x"]).data
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_remove_nans_compaction_properties(case, dev):
    x = case["x"]
    n = case["n_src"]
    holes = torch.rand(n, device=dev, generator=None) < 0.05
    first = x.data[:, 0].clone()
    first[holes] = float("nan")
    mask = torch.empty(n + 8, dtype=torch.uint8, device=dev)
    native.mask_build(first, mask, n=n, stride=1, cmp=native.CMP_NOTNAN)
    count = native.mask_count(mask, n)
    assert count == int((~holes).sum().item())
    index = native.mask_to_index(mask, n)
    assert index.numel() == count
    assert bool((index[1:] > index[:-1]).all())  # strictly increasing: numpy boolean-indexing order
    assert not bool(holes[index.long()].any())
    plan = GatherPlan(n, count, index=index.cpu().numpy())
    out = plan.apply(x)
    assert torch.equal(out.data.view(torch.int32), x.data[~holes].view(torch.int32))


def test_pointwise_round_trip_and_mask_count(case, dev):
    x = case["x"]
    n = case["n_src"]
    y = x.new_like()
    fwd = native.level_program([[(native.OP_AFFINE, 0, 1.8, 32.0)] * N_LEV], dev)
    bwd = native.level_program([[(native.OP_AFFINE_INV, 0, 1.8, 32.0)] * N_LEV], dev)
    kw = dict(n_pts=n, n_lev=N_LEV, x_pitch=x.pitch, y_pitch=y.pitch, layout=COLUMNS)
    native.pointwise_stack(x.data, y.data, prog=fwd, n_stage=1, **kw)
    native.pointwise_stack(y.data, y.data, prog=bwd, n_stage=1, **kw)
    assert torch.allclose(y.data[:, :N_LEV], x.data[:, :N_LEV], rtol=3e-7, atol=1e-4)
    pm = (torch.rand(n, device=dev) < 0.25).to(torch.uint8)
    pm = torch.cat([pm, torch.zeros(8, dtype=torch.uint8, device=dev)])
    masked = native.level_program([[(native.OP_COPY, 1, 0.0, 0.0)] * N_LEV], dev)
    native.pointwise_stack(x.data, y.data, prog=masked, n_stage=1, point_mask=pm, **kw)
    n_nan = native.reduce(y.data[:, :N_LEV].contiguous(), native.RED_NANCOUNT)
    assert n_nan == float(int(pm.sum().item()) * N_LEV)
    keep = pm[:n] == 0
    assert torch.equal(y.data[keep].view(torch.int32), x.data[keep].view(torch.int32))


def test_per_level_programs_agree_across_kernels_and_layouts(case, dev):
    """137 levels of O1280, a scale / clip / mask of its own on every level, two stages: the per-level LDS kernel (column stacks), the
    field-major rows kernel and the fused regrid epilogue (typed per-level table) evaluate the same statements — bit for bit the same
    values wherever the same numbers go in."""
    x, n = case["x"], case["n_src"]
    stages = [[(native.OP_AFFINE, 0, 1.0 + 0.001 * l, -273.15 + 0.5 * l) for l in range(N_LEV)],
              [(native.OP_CLIP, l % 2, -60.0 + 0.1 * l, 45.0) if l % 3 else (native.OP_MUL, 0, 1.0 + 1.0 / (1 + l), 0.0) for l in range(N_LEV)]]
    prog = native.level_program(stages, dev)
    pm = (torch.rand(n, device=dev) < 0.2).to(torch.uint8)
    cols = x.new_like()
    native.pointwise_stack(x.data, cols.data, n_pts=n, n_lev=N_LEV, x_pitch=x.pitch, y_pitch=cols.pitch, layout=COLUMNS, prog=prog, n_stage=2,
                           point_mask=pm)
    xf = x.to_layout(FIELDS)
    yf = xf.new_like()
    native.pointwise_stack(xf.data, yf.data, n_pts=n, n_lev=N_LEV, x_pitch=xf.pitch, y_pitch=yf.pitch, layout=FIELDS, prog=prog, n_stage=2,
                           point_mask=pm)
    back = yf.to_layout(COLUMNS)
    assert torch.equal(back.data[:, :N_LEV].contiguous().view(torch.int32), cols.data[:, :N_LEV].contiguous().view(torch.int32))
    # oracle spot check on three levels
    for l in (0, 77, 136):
        want = x.level_numpy(l)
        for stage in stages:
            op, use_mask, p0, p1 = stage[l]
            p0, p1 = np.float32(p0), np.float32(p1)
            want = oracle.rescale_forward(want, p0, p1) if op == native.OP_AFFINE else (want * p0 if op == native.OP_MUL else oracle.clip(want, p0, p1))
            if use_mask:
                want = oracle.apply_mask_values(want.copy(), pm.cpu().numpy().astype(bool))
        assert np.array_equal(cols.level_numpy(l), want, equal_nan=True), l
    # the fused epilogue of the regrid (multiply-add stage only: the direct kernel's typed table) against regrid-then-program
    madd = native.level_program([stages[0]], dev)
    fused = case["plan"].apply(x, prog=madd, n_stage=1)
    plain = case["plan"].apply(x)
    native.pointwise_stack(plain.data, plain.data, n_pts=case["n_tgt"], n_lev=N_LEV, x_pitch=plain.pitch, y_pitch=plain.pitch, layout=COLUMNS,
                           prog=madd, n_stage=1)
    assert torch.equal(fused.data[:, :N_LEV].contiguous().view(torch.int32), plain.data[:, :N_LEV].contiguous().view(torch.int32))


def test_empty_and_tiny_inputs(dev):
    """Edge cases: zero targets, one target, one level, one source point."""
    x = Stack.from_fields(np.arange(12.0).reshape(3, 4), dev=dev)
    empty = GatherPlan(4, 0, index=np.zeros((0, 1), dtype=np.int64))
    assert empty.apply(x).n_pts == 0
    one = GatherPlan(4, 1, index=np.array([[3, 0]]), weights=np.array([[0.25, 0.75]]))
    assert np.array_equal(one.apply(x).numpy()[:, 0], 0.25 * np.array([3.0, 7.0, 11.0]) + 0.75 * np.array([0.0, 4.0, 8.0]))
    single = Stack.from_fields(np.array([[5.0]]), dev=dev)
    assert np.array_equal(GatherPlan(1, 3, index=np.zeros(3, dtype=np.int64)).apply(single).numpy(), [[5.0, 5.0, 5.0]])
    assert native.mask_to_index(torch.zeros(4, dtype=torch.uint8, device=dev), 0).numel() == 0
