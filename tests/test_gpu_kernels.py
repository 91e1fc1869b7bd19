"""Parity of every libatx kernel against the CPU oracle, through the C ABI.

Bit-exact for gathers, masks, compaction and f64 arithmetic (the library is built
without FMA contraction, so f64 equals numpy / scipy bit for bit); f32
interpolation within 1e-6 relative of the oracle fed the same f32 inputs
(BASELINE.json north_star tolerance).
"""

from __future__ import annotations

import os

import numpy as np
import pytest
import torch

from anemoi_transform_amd import native
from anemoi_transform_amd.gather import GatherPlan
from anemoi_transform_amd.stack import COLUMNS, FIELDS, Stack
from oracle import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DTYPES = [(torch.float64, np.float64), (torch.float32, np.float32)]
LAYOUTS = [COLUMNS, FIELDS]
RTOL_F32 = 1e-6  # north_star: "within 1e-6 relative for float interpolation"


def make_fields(rng, n_lev, n_pts, np_dtype, nan_frac=0.0):
    x = (280.0 + 30.0 * rng.standard_normal((n_lev, n_pts))).astype(np_dtype)
    if nan_frac:
        x[rng.random((n_lev, n_pts)) < nan_frac] = np.nan
    return x


def random_ell(rng, n_src, n_tgt, k, np_dtype):
    idx = rng.integers(0, n_src, size=(n_tgt, k)).astype(np.int32)
    w = rng.random((n_tgt, k))
    w = (w / w.sum(axis=1, keepdims=True)).astype(np_dtype)
    return idx, w


def random_csr(rng, n_src, n_tgt, max_row, np_dtype, empty_rows=True):
    lengths = rng.integers(0 if empty_rows else 1, max_row + 1, size=n_tgt)
    indptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    nnz = int(indptr[-1])
    indices = rng.integers(0, n_src, size=nnz).astype(np.int32)
    data = rng.standard_normal(nnz).astype(np_dtype)
    return indptr, indices, data


def to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def assert_interp(got, want, np_dtype):
    """`want` is scipy's csr_matvec run in the stack's own width: rows are summed from 0 in index order, one rounding per
    product and per sum (libatx is built with -ffp-contract=off), so float32 is held to the same bits as float64 — tighter
    than north_star's 1e-6 relative, which is the bound against the float64 statement (tests/test_gpu_fullsize.py)."""
    assert got.dtype == want.dtype == np_dtype
    assert np.array_equal(got, want, equal_nan=True)


# ---------------------------------------------------------------------------------
# regrid
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("n_lev", [1, 3, 137])
def test_gather_nn_bit_exact(dev, tdtype, np_dtype, layout, n_lev):
    """R: regrid.py:380 — a k=1 gather copies bits, NaN payloads included."""
    rng = np.random.default_rng(1)
    n_src, n_tgt = 5000, 3333
    x = make_fields(rng, n_lev, n_src, np_dtype, nan_frac=0.05)
    # a NaN with a payload must survive
    x.view(np.uint32 if np_dtype == np.float32 else np.uint64)[0, 0] |= 0x7
    idx = rng.integers(0, n_src, size=n_tgt).astype(np.int32)
    src = Stack.from_fields(x, dev=dev, layout=layout)
    out = src.new_like(n_pts=n_tgt)
    native.regrid_ell(src.data, out.data, to_dev(idx, dev), None, n_src=n_src, n_tgt=n_tgt, k=1, n_lev=n_lev,
                      src_pitch=src.pitch, out_pitch=out.pitch, layout=layout)
    got = out.numpy()
    want = oracle.gather_nn(x, idx)
    itype = np.uint32 if np_dtype == np.float32 else np.uint64
    assert np.array_equal(got.view(itype), want.view(itype))


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 7, 8, 9, 13, 16, 33, 64])
def test_regrid_ell_matches_csr_matvec(dev, tdtype, np_dtype, layout, k):
    """R: regrid.py:310 — fixed-k rows against scipy's csr_matvec."""
    rng = np.random.default_rng(2 + k)
    n_src, n_tgt, n_lev = 4096, 2500, 19
    x = make_fields(rng, n_lev, n_src, np_dtype)
    idx, w = random_ell(rng, n_src, n_tgt, k, np_dtype)
    src = Stack.from_fields(x, dev=dev, layout=layout)
    out = src.new_like(n_pts=n_tgt)
    native.regrid_ell(src.data, out.data, to_dev(idx, dev), to_dev(w, dev), n_src=n_src, n_tgt=n_tgt, k=k,
                      n_lev=n_lev, src_pitch=src.pitch, out_pitch=out.pitch, layout=layout)
    indptr = np.arange(n_tgt + 1) * k
    want = np.stack([oracle.csr_apply(w.reshape(-1), idx.reshape(-1), indptr, (n_tgt, n_src), f) for f in x])
    assert_interp(out.numpy(), want, np_dtype)


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("layout", LAYOUTS)
def test_regrid_csr_ragged_rows(dev, tdtype, np_dtype, layout):
    """General CSR: empty rows, long rows (beyond the LDS staging cap), duplicate columns."""
    rng = np.random.default_rng(5)
    n_src, n_tgt, n_lev = 3000, 1777, 11
    x = make_fields(rng, n_lev, n_src, np_dtype)
    indptr, indices, data = random_csr(rng, n_src, n_tgt, 9, np_dtype)
    src = Stack.from_fields(x, dev=dev, layout=layout)
    out = src.new_like(n_pts=n_tgt)
    native.regrid_csr(src.data, out.data, to_dev(indptr, dev), to_dev(indices, dev), to_dev(data, dev), n_src=n_src,
                      n_tgt=n_tgt, nnz=len(indices), n_lev=n_lev, src_pitch=src.pitch, out_pitch=out.pitch, layout=layout)
    want = np.stack([oracle.csr_apply(data, indices, indptr, (n_tgt, n_src), f) for f in x])
    # sign-changing weights, empty rows, duplicate columns: scipy's own summation order, so the same bits in both widths
    assert_interp(out.numpy(), want, np_dtype)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_regrid_csr_very_long_rows(dev, layout):
    rng = np.random.default_rng(6)
    n_src, n_tgt, n_lev = 2000, 64, 5
    x = make_fields(rng, n_lev, n_src, np.float64)
    lengths = rng.integers(200, 400, size=n_tgt)
    indptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    indices = rng.integers(0, n_src, size=int(indptr[-1])).astype(np.int32)
    data = rng.random(int(indptr[-1]))
    src = Stack.from_fields(x, dev=dev, layout=layout)
    out = src.new_like(n_pts=n_tgt)
    native.regrid_csr(src.data, out.data, to_dev(indptr, dev), to_dev(indices, dev), to_dev(data, dev), n_src=n_src,
                      n_tgt=n_tgt, nnz=len(indices), n_lev=n_lev, src_pitch=src.pitch, out_pitch=out.pitch, layout=layout)
    want = np.stack([oracle.csr_apply(data, indices, indptr, (n_tgt, n_src), f) for f in x])
    assert np.array_equal(out.numpy(), want)


@pytest.mark.parametrize("n_tgt", [1, 1023, 2031, 2032, 2048, 2049, 4097, 6001])
def test_regrid_csr_striped_tiles_cover_every_row(dev, n_tgt):
    """Rows of 8 entries or more (on average) have their tiles dealt to the XCDs in stripes (xcd_stripe): whatever the tile count —
    below one group of stripes, exactly one, a tail beyond the last whole group — every row is computed once, with scipy's bits."""
    rng = np.random.default_rng(n_tgt)
    n_src, n_lev = 1500, 5  # 5 float64 levels: tiles of a few hundred targets at most, many tiles
    x = make_fields(rng, n_lev, n_src, np.float64)
    lengths = rng.integers(6, 30, size=n_tgt)
    indptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    indices = rng.integers(0, n_src, size=int(indptr[-1])).astype(np.int32)
    data = rng.random(int(indptr[-1]))
    src = Stack.from_fields(x, dev=dev, layout=COLUMNS)
    want = np.stack([oracle.csr_apply(data, indices, indptr, (n_tgt, n_src), f) for f in x])
    for tile in (0, 8):  # the heuristic's tile and a forced small one (many tiles)
        native.set_tuning(tile)
        try:
            out = src.new_like(n_pts=n_tgt)
            out.data.fill_(float("nan"))
            native.regrid_csr(src.data, out.data, to_dev(indptr, dev), to_dev(indices, dev), to_dev(data, dev), n_src=n_src,
                              n_tgt=n_tgt, nnz=len(indices), n_lev=n_lev, src_pitch=src.pitch, out_pitch=out.pitch, layout=COLUMNS)
        finally:
            native.set_tuning(0)
        assert np.array_equal(out.numpy(), want), tile


def test_regrid_unaligned_columns_take_scalar_path(dev):
    """A columns stack whose pitch is not a 16-byte multiple still regrids correctly."""
    rng = np.random.default_rng(7)
    n_src, n_tgt, n_lev = 1000, 700, 5
    x = make_fields(rng, n_lev, n_src, np.float32)
    idx, w = random_ell(rng, n_src, n_tgt, 4, np.float32)
    src_t = torch.zeros((n_src, 5), dtype=torch.float32, device=dev)
    src_t.copy_(to_dev(x.T, dev))
    out_t = torch.empty((n_tgt, 7), dtype=torch.float32, device=dev)
    native.regrid_ell(src_t, out_t, to_dev(idx, dev), to_dev(w, dev), n_src=n_src, n_tgt=n_tgt, k=4, n_lev=n_lev,
                      src_pitch=5, out_pitch=7, layout=COLUMNS)
    indptr = np.arange(n_tgt + 1) * 4
    want = np.stack([oracle.csr_apply(w.reshape(-1), idx.reshape(-1), indptr, (n_tgt, n_src), f) for f in x])
    np.testing.assert_allclose(out_t[:, :n_lev].T.cpu().numpy(), want, rtol=RTOL_F32)


def test_regrid_fused_epilogue(dev):
    """regrid -> orog_to_z -> rescale fused in the gather epilogue equals the chained oracle."""
    rng = np.random.default_rng(8)
    n_src, n_tgt, n_lev = 2048, 1500, 6
    x = make_fields(rng, n_lev, n_src, np.float64)
    idx, w = random_ell(rng, n_src, n_tgt, 4, np.float64)
    tmask = rng.random(n_tgt) < 0.2
    stage0 = [(native.OP_MUL, 0, oracle.G, 0.0) if l == 1 else (native.OP_COPY, 0, 0, 0) for l in range(n_lev)]
    stage1 = [(native.OP_AFFINE, 1 if l == 3 else 0, 1.0, -273.15) if l in (0, 3) else (native.OP_COPY, 0, 0, 0) for l in range(n_lev)]
    for layout in LAYOUTS:
        src = Stack.from_fields(x, dev=dev, layout=layout)
        out = src.new_like(n_pts=n_tgt)
        prog = native.level_program([stage0, stage1], dev)
        native.regrid_ell(src.data, out.data, to_dev(idx, dev), to_dev(w, dev), n_src=n_src, n_tgt=n_tgt, k=4,
                          n_lev=n_lev, src_pitch=src.pitch, out_pitch=out.pitch, layout=layout, prog=prog, n_stage=2,
                          tgt_mask=to_dev(tmask.astype(np.uint8), dev))
        indptr = np.arange(n_tgt + 1) * 4
        want = np.stack([oracle.csr_apply(w.reshape(-1), idx.reshape(-1), indptr, (n_tgt, n_src), f) for f in x])
        want[1] = oracle.orog_to_z(want[1])
        want[0] = oracle.rescale_forward(want[0], 1.0, -273.15)
        want[3] = oracle.rescale_forward(want[3], 1.0, -273.15)
        want[3][tmask] = np.nan
        assert np.array_equal(out.numpy(), want, equal_nan=True)


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("k,padded", [(1, False), (4, False), (3, True)])
@pytest.mark.parametrize("program", ["uniform", "per_vector", "three_pieces", "mixed_madd", "mixed_vectors", "two_pieces_general", "masked",
                                     "masked_uniform", "masked_3_stages", "scale_per_level"])
def test_fused_epilogue_kernel_variants_agree(dev, tdtype, np_dtype, k, padded, program):
    """The epilogue reaches the gather by three routes — operators by value in the kernel arguments (uniform programs seen
    through host_prog), the host-built per-vector table (vec_prog), the tiled kernel's LDS table (neither companion) — and
    all three must give the bits of the oracle's chain `csr @ x` then the per-level statements."""
    rng = np.random.default_rng(31)
    n_src, n_tgt, n_lev = 3000, 2111, 21  # 21 levels: a ragged last vector for both dtypes
    x = make_fields(rng, n_lev, n_src, np_dtype)
    idx, w = random_ell(rng, n_src, n_tgt, k, np_dtype)
    if padded:
        drop = rng.random(idx.shape) < 0.2
        drop[:, 0] = False
        idx = np.where(drop, -1, idx).astype(np.int32)
        w = np.where(drop, 0.0, w).astype(np_dtype)
    mul, aff, cp = (native.OP_MUL, 0, oracle.G, 0.0), (native.OP_AFFINE, 0, 1.0, -273.15), (native.OP_COPY, 0, 0.0, 0.0)
    if program == "uniform":  # every level the same two operators: eligible for the by-value kernel
        stages = [[mul] * n_lev, [aff] * n_lev]
    elif program == "per_vector":  # operators change at vector boundaries of BOTH dtypes (multiples of 4 levels)
        stages = [[mul if l < 8 else cp for l in range(n_lev)], [aff if l >= 12 else cp for l in range(n_lev)]]
    elif program == "three_pieces":  # three runs of levels: beyond the two pieces that travel by value -> per-vector table
        stages = [[mul if l < 4 else (cp if l < 12 else aff) for l in range(n_lev)]]
    elif program == "two_pieces_general":  # two pieces, operators outside the multiply-add family (by value: both evaluated, one kept)
        stages = [[(native.OP_CLIP, 0, 250.0, 300.0) if l < 16 else (native.OP_AFFINE_INV, 0, 2.0, 1.0) for l in range(n_lev)],
                  [(native.OP_LOG, 0, 0.0, 0.0) if l < 4 else (native.OP_DIV, 0, 3.0, 0.0) for l in range(n_lev)]]
    elif program == "masked_uniform":  # convert everywhere, then apply_mask everywhere: by value, with the mask
        stages = [[aff] * n_lev, [(native.OP_COPY, 1, 0.0, 0.0)] * n_lev]
    elif program == "mixed_madd":  # operators differ inside a vector, all of the multiply-add family (direct kernel, per-level path)
        stages = [[mul if l % 3 == 0 else cp for l in range(n_lev)], [aff if l % 2 else cp for l in range(n_lev)]]
    elif program == "scale_per_level":  # normalisation per level: every level its own parameters (the typed per-level part of vec_prog)
        stages = [[(native.OP_MUL, 0, 1.0 + 0.01 * l, 0.0) for l in range(n_lev)], [(native.OP_AFFINE, 0, 1.0 / (1 + l), -0.37 * l) for l in range(n_lev)]]
    elif program == "masked_3_stages":  # BASELINE config 5 with apply_mask: orog_to_z on one level, convert on the others, mask on some
        stages = [[cp] * (n_lev - 1) + [mul], [aff] * (n_lev - 1) + [cp], [(native.OP_COPY, 1 if l % 5 else 0, 0.0, 0.0) for l in range(n_lev)]]
    elif program == "mixed_vectors":  # operators differ inside a vector, general operators (tiled kernel)
        stages = [[mul if l % 3 == 0 else cp for l in range(n_lev)], [aff if l % 2 else (native.OP_CLIP, 0, 250.0, 300.0) for l in range(n_lev)]]
    else:
        stages = [[(native.OP_AFFINE, 1 if l in (2, 20) else 0, 2.0, 1.0) for l in range(n_lev)]]
    tmask = rng.random(n_tgt) < 0.3
    tmask_d = to_dev(tmask.astype(np.uint8), dev) if program.startswith("masked") else None
    src = Stack.from_fields(x, dev=dev)
    idx_d, w_d = to_dev(idx, dev), (to_dev(w, dev) if (k > 1 or padded) else None)

    def run(strip):
        prog = native.level_program(stages, dev)
        for name in strip:
            if name == "vec_prog":
                prog.vec_prog = {}
            else:
                prog.host_prog = None
        out = src.new_like(n_pts=n_tgt)
        native.regrid_ell(src.data, out.data, idx_d, w_d, n_src=n_src, n_tgt=n_tgt, k=k, n_lev=n_lev, src_pitch=src.pitch,
                          out_pitch=out.pitch, layout=COLUMNS, prog=prog, n_stage=len(stages), tgt_mask=tmask_d, padded=padded)
        return out.numpy()

    full, table_only, tiled = run(()), run(("host_prog",)), run(("host_prog", "vec_prog"))
    # oracle chain
    if k == 1 and not padded:
        want = np.stack([oracle.gather_nn(f, idx[:, 0]) for f in x])
    else:
        present = idx >= 0
        lengths = present.sum(axis=1)
        indptr = np.concatenate([[0], np.cumsum(lengths)])
        want = np.stack([oracle.csr_apply(w[present], idx[present], indptr, (n_tgt, n_src), f) for f in x])
    table = native.LEVEL_OP_DTYPE
    for stage in stages:
        for l, (op, use_mask, p0, p1) in enumerate(stage):
            entry = np.zeros((), dtype=table)
            entry["op"], entry["use_mask"], entry["p0"], entry["p1"] = op, use_mask, p0, p1
            import native_double

            want[l] = native_double._apply_op(entry, want[l], tmask if use_mask else None)
    libm = any(op in (native.OP_LOG, native.OP_EXP) for stage in stages for (op, _, _, _) in stage)  # ocml vs libm: a few ulp
    for got in (full, table_only, tiled):
        if (np_dtype == np.float64 or (k == 1 and not padded)) and not libm:
            assert np.array_equal(got, want, equal_nan=True)
        else:
            np.testing.assert_allclose(got, want, rtol=RTOL_F32 if np_dtype == np.float32 else 1e-14, atol=1e-4, equal_nan=True)
    assert np.array_equal(full, tiled, equal_nan=True) and np.array_equal(table_only, tiled, equal_nan=True)  # same arithmetic, same bits


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("k,padded,program", [(1, False, None), (4, False, None), (3, True, None), (4, False, "masked_uniform"), (4, False, "per_level"),
                                              (7, False, None), (7, False, "masked_uniform"), (13, False, None), (13, False, "per_level"),
                                              (11, True, "masked_uniform")])
def test_ordered_traversal_gives_the_natural_orders_bits(dev, tdtype, np_dtype, k, padded, program):
    """`atx_regrid_ell_ordered`: the tables permuted into a visiting order + `tgt_rows` give, bit for bit, what the un-permuted
    tables give — direct kernel (k <= 4, every epilogue route), tiled kernel (k = 7 and a forced tile), one stack and a batch of
    three, the target mask indexed by OUTPUT row."""
    rng = np.random.default_rng(31)
    n_src, n_tgt, n_lev = 5000, 3001, 21
    x = [make_fields(rng, n_lev, n_src, np_dtype) for _ in range(3)]
    idx, w = random_ell(rng, n_src, n_tgt, k, np_dtype)
    if padded:
        drop = rng.random((n_tgt, k)) < 0.25
        drop[:, 0] = False
        idx = np.where(drop, -1, idx).astype(np.int32)
        w = np.where(drop, 0.0, w).astype(np_dtype)
    aff, msk = (native.OP_AFFINE, 0, 1.0, -273.15), (native.OP_COPY, 1, 0.0, 0.0)
    stages = None
    if program == "masked_uniform":
        stages = [[aff] * n_lev, [msk] * n_lev]
    elif program == "per_level":
        stages = [[(native.OP_AFFINE, 0, 1.0 + 0.01 * l, float(l)) for l in range(n_lev)]]
    tmask = to_dev((rng.random(n_tgt) < 0.3).astype(np.uint8), dev) if program == "masked_uniform" else None
    order = rng.permutation(n_tgt).astype(np.int32)
    srcs = [Stack.from_fields(f, dev=dev) for f in x]
    weighted = k > 1 or padded
    tables = {"natural": (to_dev(idx, dev), to_dev(w, dev) if weighted else None, None),
              "ordered": (to_dev(idx[order], dev), to_dev(w[order], dev) if weighted else None, to_dev(order, dev))}

    def run(which, n_stack, tile=0):
        i_d, w_d, rows = tables[which]
        prog = native.level_program(stages, dev) if stages else None
        outs = [srcs[0].new_like(n_pts=n_tgt) for _ in range(n_stack)]
        native.set_tuning(tile)
        try:
            native.regrid_ell_batch([s.data for s in srcs[:n_stack]], [o.data for o in outs], i_d, w_d, n_src=n_src, n_tgt=n_tgt, k=k, n_lev=n_lev,
                                    src_pitch=srcs[0].pitch, out_pitch=outs[0].pitch, layout=COLUMNS, prog=prog, n_stage=len(stages) if stages else 0,
                                    tgt_mask=tmask, padded=padded, tgt_rows=rows)
        finally:
            native.set_tuning(0)
        return [o.numpy() for o in outs]

    for n_stack in (1, 3):
        for tile in (0, 16):
            want = run("natural", n_stack, tile)
            got = run("ordered", n_stack, tile)
            assert all(np.array_equal(g, v, equal_nan=True) for g, v in zip(got, want)), (n_stack, tile)
    # field-major stacks have no ordered form: refused, not ignored
    f = srcs[0].to_layout(FIELDS)
    with pytest.raises(NotImplementedError):
        i_d, w_d, rows = tables["ordered"]
        native.regrid_ell(f.data, f.new_like(n_pts=n_tgt).data, i_d, w_d, n_src=n_src, n_tgt=n_tgt, k=k, n_lev=n_lev, src_pitch=f.pitch,
                          out_pitch=f.new_like(n_pts=n_tgt).pitch, layout=FIELDS, padded=padded, tgt_rows=rows)


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("with_prog", [False, True])
def test_ordered_csr_traversal_gives_the_natural_orders_bits(dev, tdtype, np_dtype, with_prog):
    """`atx_regrid_csr_ordered` through `GatherPlan.order_targets`: the CSR matrix with its rows permuted into a visiting order +
    `tgt_rows` equals the un-permuted matrix bit for bit (ragged rows, empty rows, a masked epilogue indexed by output row)."""
    from anemoi_transform_amd.gather import GatherPlan

    rng = np.random.default_rng(37)
    n_src, n_tgt, n_lev = 4000, 2500, 19
    indptr, indices, data = random_csr(rng, n_src, n_tgt, 12, np_dtype)
    x = Stack.from_fields(make_fields(rng, n_lev, n_src, np_dtype), dev=dev)
    natural = GatherPlan(n_src, n_tgt, csr=(data, indices, indptr))
    ordered = GatherPlan(n_src, n_tgt, csr=(data, indices, indptr)).order_targets(rng.permutation(n_tgt))
    kw = {}
    if with_prog:
        stages = [[(native.OP_AFFINE, 0, 1.0, -273.15)] * n_lev, [(native.OP_COPY, 1, 0.0, 0.0)] * n_lev]
        kw = dict(prog=native.level_program(stages, dev), n_stage=2, tgt_mask=to_dev((rng.random(n_tgt) < 0.3).astype(np.uint8), dev))
    want = natural.apply(x, **kw).numpy()
    assert np.array_equal(ordered.apply(x, **kw).numpy(), want, equal_nan=True)
    native.set_tuning(8)
    try:
        assert np.array_equal(ordered.apply(x, **kw).numpy(), want, equal_nan=True)
    finally:
        native.set_tuning(0)
    lo, hi = ordered.shard_range(2, 3)
    assert np.array_equal(ordered.shard(2, 3).apply(x).numpy(), natural.apply(x).numpy()[:, lo:hi])
    assert np.array_equal(ordered.apply(x.to_layout(FIELDS)).numpy(), natural.apply(x).numpy())  # field-major: order ignored


def test_check_indices(dev):
    idx = np.array([0, 5, 9, 10, -1, 3], dtype=np.int32)
    assert native.check_indices(to_dev(idx, dev), 10) == 2
    assert native.check_indices(to_dev(idx[:3], dev), 10) == 0


def test_argument_errors_map_to_reference_exceptions(dev):
    t = torch.zeros((4, 4), dtype=torch.float32, device=dev)
    i = torch.zeros(4, dtype=torch.int32, device=dev)
    with pytest.raises(AssertionError):  # pitch < n_lev: shape mismatch (R: regrid.py:377-378)
        native.regrid_ell(t, t.clone(), i, None, n_src=4, n_tgt=4, k=1, n_lev=8, src_pitch=4, out_pitch=4, layout=COLUMNS)
    with pytest.raises(ValueError):
        native.regrid_ell(t, t.clone(), i, None, n_src=4, n_tgt=4, k=3, n_lev=4, src_pitch=4, out_pitch=4, layout=COLUMNS)
    with pytest.raises(RuntimeError):  # CPU tensors are refused: no fallback
        native.regrid_ell(t.cpu(), t.cpu(), i.cpu(), None, n_src=4, n_tgt=4, k=1, n_lev=4, src_pitch=4, out_pitch=4, layout=COLUMNS)


@pytest.mark.parametrize("workspace", [True, False])
def test_reduce_routes_with_and_without_a_workspace(dev, monkeypatch, workspace):
    """`atx_reduce[_stack]` given a workspace (the default of native.py): partials, a one-workgroup finish, the result written
    straight into a pinned host cell; without one: per-workgroup atomics on a device cell."""
    monkeypatch.setattr(native, "_REDUCE_TICKET", workspace)
    rng = np.random.default_rng(18)
    for np_dtype in (np.float64, np.float32):
        # empty input: the identities, through the workspace's finish as through reduce_init (np.min of nothing raises; the library
        # answers +inf / -inf / 0 so that callers can combine ranges)
        empty = torch.empty(0, dtype=torch.float64 if np_dtype == np.float64 else torch.float32, device=dev)
        assert native.reduce(empty, native.RED_MINMAX) == (float("inf"), float("-inf"))
        assert native.reduce(empty, native.RED_MIN) == float("inf") and native.reduce(empty, native.RED_MAX) == float("-inf")
        assert native.reduce(empty, native.RED_NANCOUNT) == 0.0
        # 3: nothing but a tail; 5_000_001: whole vectors and a tail (round 3 sent tails through atomics on the caller's pinned cell;
        # with a workspace they are one more partial now); an unaligned base takes the scalar kernels, MINMAX in two passes
        for n in (3, 4, 4096, 1_000_000, 5_000_001):
            x = make_fields(rng, 1, n, np_dtype)[0]
            xd = to_dev(x, dev)
            for _ in range(3):  # the workspace is reused call after call
                assert native.reduce(xd, native.RED_MINMAX) == (float(x.min()), float(x.max()))
            assert native.reduce(xd, native.RED_MIN) == float(x.min()) and native.reduce(xd, native.RED_NANCOUNT) == 0.0
            shifted = to_dev(np.concatenate([[np_dtype(1e9)], x]), dev)[1:]  # base off the 16-byte boundary; the 1e9 in front must not be seen
            assert native.reduce(shifted, native.RED_MINMAX) == (float(x.min()), float(x.max()))
            x[n // 2] = np.nan
            xd = to_dev(x, dev)
            assert all(np.isnan(v) for v in native.reduce(xd, native.RED_MINMAX)) and native.reduce(xd, native.RED_NANCOUNT) == 1.0
            x[n // 2], x[-1] = 0.0, np.nan  # the NaN in the tail
            assert native.reduce(to_dev(x, dev), native.RED_NANCOUNT) == 1.0 and np.isnan(native.reduce(to_dev(x, dev), native.RED_MAX))
        z = make_fields(rng, 37, 20011, np_dtype)
        st = Stack.from_fields(z, dev=dev)
        kw = dict(n_pts=st.n_pts, n_lev=st.n_lev, pitch=st.pitch, layout=COLUMNS)
        assert native.reduce_stack(st.data, native.RED_MINMAX, **kw) == (float(z.min()), float(z.max()))
        assert native.reduce_stack(st.data, native.RED_MAX, **kw) == float(z.max())


# ---------------------------------------------------------------------------------
# per-point programs
# ---------------------------------------------------------------------------------
def _run_prog(dev, x, layout, stages, mask=None, in_place=False):
    n_lev, n_pts = x.shape
    src = Stack.from_fields(x, dev=dev, layout=layout)
    out = src if in_place else src.new_like()
    prog = native.level_program(stages, dev)
    native.pointwise_stack(src.data, out.data, n_pts=n_pts, n_lev=n_lev, x_pitch=src.pitch, y_pitch=out.pitch,
                           layout=layout, prog=prog, n_stage=len(stages),
                           point_mask=None if mask is None else to_dev(mask.astype(np.uint8), dev))
    return out.numpy()


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("in_place", [False, True])
def test_pointwise_ops_bit_exact(dev, tdtype, np_dtype, layout, in_place):
    rng = np.random.default_rng(11)
    n_lev, n_pts = 10, 4099
    x = make_fields(rng, n_lev, n_pts, np_dtype, nan_frac=0.03)
    x[4] = np.abs(x[4]) * 1e-2  # exp / log inputs
    x[5] = np.abs(x[5]) + 1.0
    x[0, :8] = [-0.0, 0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 273.15]
    sc, off = np_dtype(1.8), np_dtype(-459.67)
    stage = [
        (native.OP_AFFINE, 0, 1.8, -459.67),
        (native.OP_AFFINE_INV, 0, 1.8, -459.67),
        (native.OP_MUL, 0, oracle.G, 0),
        (native.OP_DIV, 0, oracle.G, 0),
        (native.OP_EXP, 0, 0, 0),
        (native.OP_LOG, 0, 0, 0),
        (native.OP_CLIP, 0, 250.0, 300.0),
        (native.OP_CLIP, 0, float("nan"), 280.0),
        (native.OP_IMPUTE_NAN, 0, -1.0, 0),
        (native.OP_COPY, 0, 0, 0),
    ]
    got = _run_prog(dev, x, layout, [stage], in_place=in_place)
    g = np_dtype(oracle.G)
    want = [
        oracle.rescale_forward(x[0], sc, off),
        oracle.rescale_backward(x[1], sc, off),
        x[2] * g,
        x[3] / g,
        oracle.lnsp_to_sp(x[4]),
        oracle.sp_to_lnsp(x[5]),
        oracle.clip(x[6], np_dtype(250.0), np_dtype(300.0)),
        oracle.clip(x[7], None, np_dtype(280.0)),
        oracle.impute_nans(x[8], np_dtype(-1.0)),
        x[9],
    ]
    for l in (0, 1, 2, 3, 6, 7, 8, 9):  # one-rounding arithmetic: bit-exact
        assert want[l].dtype == np_dtype, (l, want[l].dtype)
        assert np.array_equal(got[l], want[l], equal_nan=True), f"level {l}"
        # signed zeros must match too (x*g keeps -0.0, x*1+0 does not)
        finite = ~np.isnan(want[l])
        assert np.array_equal(np.signbit(got[l][finite]), np.signbit(want[l][finite])), f"level {l} sign"
    for l in (4, 5):  # libm vs ocml: a few ulp
        np.testing.assert_allclose(got[l], want[l], rtol=1e-6 if np_dtype == np.float32 else 1e-14, equal_nan=True)


def test_float64_log_within_one_ulp_of_numpy(dev):
    """ATX_OP_LOG in float64 is evaluated by the library's own argument reduction + polynomial (atx_common.hpp: atx_log) instead of
    the device library's log: at most 1 ulp from numpy over the whole positive range, exact at 1, IEEE results at 0, negatives, inf, NaN
    and subnormals — through every kernel that can run the operator (by value, per-level tables, fused regrid epilogue)."""
    rng = np.random.default_rng(77)
    n = 1 << 18
    cases = {
        "surface pressure (Pa)": rng.uniform(3.0e4, 1.1e5, n),
        "around 1": 1.0 + rng.uniform(-0.5, 1.0, n),
        "next to 1": 1.0 + rng.uniform(-1e-9, 1e-9, n),
        "every magnitude": 10.0 ** rng.uniform(-307, 308, n),
        "subnormal": rng.uniform(5e-324, 2.2e-308, n),
    }
    # where the reduction switches: powers of two and 2^k sqrt(1/2) (the mantissa is doubled below sqrt(1/2)), a few ulp either side, every exponent
    k = np.arange(-1074, 1024, dtype=np.float64)
    near = []
    for base in (np.ldexp(1.0, k.astype(int)), np.ldexp(np.sqrt(0.5), k[k > -1073].astype(int))):
        for step in (-3, -2, -1, 0, 1, 2, 3):
            v = base.copy()
            for _ in range(abs(step)):
                v = np.nextafter(v, np.inf if step > 0 else 0.0)
            near.append(v)
    near = np.concatenate(near)
    cases["reduction boundaries"] = np.resize(near[near > 0], n)
    special = np.array([1.0, 0.0, -0.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 2.0, 0.5, np.e])

    def run(x, per_level=False):
        st = Stack.from_fields(np.stack([x, x[::-1].copy()]), dev=dev)
        out = st.new_like()
        second = (native.OP_LOG, 0, 1.0, 0.0) if per_level else (native.OP_LOG, 0, 0.0, 0.0)  # different parameters: not uniform over the levels
        prog = native.level_program([[(native.OP_LOG, 0, 0.0, 0.0), second]], dev)
        native.pointwise_stack(st.data, out.data, n_pts=len(x), n_lev=2, x_pitch=st.pitch, y_pitch=out.pitch, layout=COLUMNS, prog=prog, n_stage=1)
        got = out.numpy()
        assert np.array_equal(got[0], got[1][::-1], equal_nan=True)
        return got[0]

    for name, x in cases.items():
        want = np.log(x)
        for per_level in (False, True):
            got = run(x, per_level)
            err = np.abs(got - want) / np.spacing(np.abs(want))
            assert float(err.max()) <= 1.0, (name, per_level, float(err.max()))
    with np.errstate(all="ignore"):
        want = np.log(special)
    for per_level in (False, True):
        got = run(special, per_level)
        assert np.array_equal(got[:7], want[:7], equal_nan=True), (got[:7], want[:7])
        assert np.all(np.abs(got[7:] - want[7:]) <= np.spacing(np.abs(want[7:])))
    assert run(special)[0] == 0.0  # log(1) is exactly 0


def test_float64_exp_within_one_ulp_of_numpy(dev):
    """ATX_OP_EXP in float64 (lnsp_to_sp, R: lnsp_to_sp.py:47) is evaluated by the library's own routine (atx_common.hpp: atx_exp —
    argument reduction x = k ln2 + r with fused multiply-adds, exp(r) = 1 + r (1 + r q(r)) with a degree-9 q, exact scaling by
    v_ldexp_f64; 18 VALU instructions per element against the device library's 22): at most 1 ulp from numpy on the ranges the filter
    sees and beyond, exact at 0, IEEE results for overflow, underflow to subnormals and zero, inf and NaN; through every kernel that can
    run the operator, and log(exp(x)) / exp(log(x)) round trips through a two-stage program."""
    rng = np.random.default_rng(78)
    n = 1 << 18
    cases = {
        "ln of surface pressure": rng.uniform(np.log(3.0e4), np.log(1.1e5), n),
        "around 0": rng.uniform(-1.0, 1.0, n),
        "next to 0": rng.uniform(-1e-9, 1e-9, n),
        "every magnitude": rng.uniform(-708.0, 709.0, n),
        "into the subnormals": rng.uniform(-745.0, -708.0, n),
    }
    # where the reduction switches: x log2(e) next to a half-integer (k = rint(...) ties) and x next to a multiple of ln 2 (r next to 0)
    m = np.arange(-1074, 1023, dtype=np.float64)  # (results stay finite and above the smallest subnormal: the ulp distance is defined)
    near = []
    for base in ((m + 0.5) * np.log(2.0), m * np.log(2.0)):
        for step in (-2, -1, 0, 1, 2):
            v = base.copy()
            for _ in range(abs(step)):
                v = np.nextafter(v, np.inf if step > 0 else -np.inf)
            near.append(v)
    cases["reduction boundaries"] = np.resize(np.concatenate(near), n)
    # (beyond +-1100 the routine clamps its argument before reducing it — k = rint(x log2 e) must fit an int: every magnitude between
    # there and the largest double, and the points either side of the clamp, must still give +inf / 0; the humidity operators can produce
    # such arguments next to the poles of their Magnus quotients, t close to 32.19 or -0.7)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 710.0, -746.0, 1.0, -1.0, 709.782712893384, -745.1332191019411, 1e-300, 0.5,
                        1099.0, -1099.0, 1100.0, -1100.0, np.nextafter(1100.0, np.inf), np.nextafter(-1100.0, -np.inf), 1e5, -1e5, 2.0e9, -2.0e9,
                        3.1e9, -3.1e9, 1e12, -1e12, 2.0 ** 50, -(2.0 ** 50), 2.0 ** 51, -(2.0 ** 51), 1e300, -1e300, np.finfo(np.float64).max,
                        -np.finfo(np.float64).max])

    def run(x, per_level=False, stages=(native.OP_EXP,)):
        st = Stack.from_fields(np.stack([x, x[::-1].copy()]), dev=dev)
        out = st.new_like()
        rows = [[(op, 0, 0.0, 0.0), (op, 0, 1.0 if per_level else 0.0, 0.0)] for op in stages]  # different parameters: not uniform over the levels
        prog = native.level_program(rows, dev)
        native.pointwise_stack(st.data, out.data, n_pts=len(x), n_lev=2, x_pitch=st.pitch, y_pitch=out.pitch, layout=COLUMNS, prog=prog, n_stage=len(stages))
        got = out.numpy()
        assert np.array_equal(got[0], got[1][::-1], equal_nan=True)
        return got[0]

    for name, x in cases.items():
        with np.errstate(all="ignore"):
            want = np.exp(x)
        for per_level in (False, True):
            got = run(x, per_level)
            err = np.abs(got - want) / np.spacing(np.abs(want))
            assert float(err.max()) <= 1.0, (name, per_level, float(err.max()))
    with np.errstate(all="ignore"):
        want = np.exp(special)
    for per_level in (False, True):
        got = run(special, per_level)
        assert np.array_equal(got[:7], want[:7], equal_nan=True), (got[:7], want[:7])  # 1, 1, inf, 0, NaN, inf, 0
        decided = ~np.isfinite(want) | (want == 0.0)  # overflow / total underflow: exactly numpy's +inf / 0 (NaN: the NaN)
        assert np.array_equal(got[decided], want[decided], equal_nan=True), (special[decided], got[decided])
        live = ~decided
        assert np.all(np.abs(got[live] - want[live]) <= np.spacing(np.abs(want[live])))
    # sp_to_lnsp | lnsp_to_sp and back as ONE two-stage launch (what tools/kernel_bench.py times): each stage within 1 ulp of numpy's
    # function of the previous stage's OWN result is what the chain can promise, i.e. 1 ulp + the conditioning of the second function
    p = rng.uniform(3.0e4, 1.1e5, n)
    got = run(p, stages=(native.OP_LOG, native.OP_EXP))
    assert float(np.max(np.abs(got - p) / p)) <= 13 * 2.3e-16  # |d exp| = ln p (<= 11.6) ulps of the logarithm, + 1
    lnp = np.log(p)
    got = run(lnp, stages=(native.OP_EXP, native.OP_LOG))
    assert float(np.max(np.abs(got - lnp) / np.spacing(lnp))) <= 2.0


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("in_place", [False, True])
@pytest.mark.parametrize("program", ["uniform_1", "uniform_4", "uniform_5", "two_pieces", "two_pieces_one_idle", "masked_uniform", "masked_piece",
                                     "transcendental", "per_level"])
def test_pointwise_kernel_routes_agree(dev, tdtype, np_dtype, in_place, program):
    """The per-point launch takes one of three kernels on a tight column stack — operators by value (programs uniform over the
    levels, or two pieces split on a 16-byte boundary; needs the program's host copy), the per-vector table, the chunked kernel —
    and all three must give the oracle's values and each other's bits.  n_lev = 24: vectors of 4 (f32) and 2 (f64) levels."""
    import native_double

    rng = np.random.default_rng(23)
    n_lev, n_pts = 24, 5003
    x = make_fields(rng, n_lev, n_pts, np_dtype, nan_frac=0.02)
    x[:4] = np.abs(x[:4]) + 1.0  # log inputs
    mul, aff, cp = (native.OP_MUL, 0, oracle.G, 0.0), (native.OP_AFFINE, 0, 1.0, -273.15), (native.OP_COPY, 0, 0.0, 0.0)
    clip, inv, imp = (native.OP_CLIP, 0, -20.0, 40.0), (native.OP_AFFINE_INV, 0, 2.0, 1.0), (native.OP_IMPUTE_NAN, 0, -9.0, 0.0)
    msk = (native.OP_COPY, 1, 0.0, 0.0)
    if program == "uniform_1":
        stages = [[aff] * n_lev]
    elif program == "uniform_4":  # the most stages that travel by value
        stages = [[mul] * n_lev, [aff] * n_lev, [clip] * n_lev, [imp] * n_lev]
    elif program == "uniform_5":  # one stage more: chunked kernel whatever the companions
        stages = [[mul] * n_lev, [aff] * n_lev, [clip] * n_lev, [imp] * n_lev, [inv] * n_lev]
    elif program == "two_pieces":
        stages = [[aff if l < 8 else mul for l in range(n_lev)], [clip if l < 16 else inv for l in range(n_lev)]]
    elif program == "two_pieces_one_idle":  # in place the idle piece must not be touched (and out of place it must be copied)
        stages = [[cp if l < 12 else aff for l in range(n_lev)], [cp if l < 12 else (native.OP_DIV, 0, 3.0, 0.0) for l in range(n_lev)]]
    elif program == "masked_uniform":
        stages = [[aff] * n_lev, [msk] * n_lev]
    elif program == "masked_piece":  # the mask on the second piece only
        stages = [[cp if l < 8 else msk for l in range(n_lev)]]
    elif program == "transcendental":
        stages = [[(native.OP_LOG, 0, 0.0, 0.0) if l < 4 else aff for l in range(n_lev)], [(native.OP_EXP, 0, 0.0, 0.0) if l < 4 else cp for l in range(n_lev)]]
    else:  # operators differ inside a vector
        stages = [[aff if l % 3 else mul for l in range(n_lev)]]
    pmask = rng.random(n_pts) < 0.3
    uses_mask = any(e[1] for st in stages for e in st)
    pm_d = to_dev(pmask.astype(np.uint8), dev) if uses_mask else None

    def run(strip):
        prog = native.level_program(stages, dev)
        for name in strip:
            if name == "vec_prog":
                prog.vec_prog = {}
            else:
                prog.host_prog = None
        src = Stack.from_fields(x, dev=dev)
        out = src if in_place else src.new_like()
        native.pointwise_stack(src.data, out.data, n_pts=n_pts, n_lev=n_lev, x_pitch=src.pitch, y_pitch=out.pitch, layout=COLUMNS, prog=prog,
                               n_stage=len(stages), point_mask=pm_d)
        return out.numpy()

    by_value, table, chunked = run(()), run(("host_prog",)), run(("host_prog", "vec_prog"))
    want = x.copy()
    for stage in stages:
        for l, (op, use_mask, p0, p1) in enumerate(stage):
            entry = np.zeros((), dtype=native.LEVEL_OP_DTYPE)
            entry["op"], entry["use_mask"], entry["p0"], entry["p1"] = op, use_mask, p0, p1
            want[l] = native_double._apply_op(entry, want[l], pmask if use_mask else None)
    for got in (by_value, table, chunked):
        if program == "transcendental":  # ocml vs libm: a few ulp per call, and exp(log(x)) chains two of them
            np.testing.assert_allclose(got, want, rtol=4 * RTOL_F32 if np_dtype == np.float32 else 1e-14, equal_nan=True)
        else:
            assert np.array_equal(got, want, equal_nan=True)
    assert np.array_equal(by_value, chunked, equal_nan=True) and np.array_equal(table, chunked, equal_nan=True)


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("selected", [[5], [0, 77, 136], [3, 4, 5, 6, 7, 8], list(range(0, 137, 7)), list(range(0, 137, 2))])
def test_pointwise_in_place_with_few_active_levels(dev, tdtype, np_dtype, selected):
    """In place, a program that leaves most levels alone visits only the columns of the active ones (sparse kernel: up to 16
    vector columns and a third of the stack; beyond that the sweeping kernels): same stack as the out-of-place call,
    untouched levels bit-identical to the input, with and without the host-side companions of the program."""
    rng = np.random.default_rng(5)
    n_pts, n_lev = 7001, 137
    x = make_fields(rng, n_lev, n_pts, np_dtype, nan_frac=0.01)
    mask = rng.random(n_pts) < 0.25
    cp = (native.OP_COPY, 0, 0.0, 0.0)
    stages = [[(native.OP_AFFINE, 0, 2.0, -1.0) if l in selected else cp for l in range(n_lev)],
              [(native.OP_CLIP, 1 if l == selected[0] else 0, 100.0, 600.0) if l in selected else cp for l in range(n_lev)]]
    mask_d = to_dev(mask.astype(np.uint8), dev)
    src = Stack.from_fields(x, dev=dev)
    want = src.new_like()
    kw = dict(n_pts=n_pts, n_lev=n_lev, x_pitch=src.pitch, y_pitch=want.pitch, layout=COLUMNS, n_stage=2, point_mask=mask_d)
    native.pointwise_stack(src.data, want.data, prog=native.level_program(stages, dev), **kw)
    expect = x.copy()
    for l in selected:
        expect[l] = oracle.clip(oracle.rescale_forward(x[l], np_dtype(2.0), np_dtype(-1.0)), np_dtype(100.0), np_dtype(600.0))
    expect[selected[0]][mask] = np.nan
    assert np.array_equal(want.numpy(), expect, equal_nan=True)
    for strip in (False, True):
        y = Stack(src.data.clone(), n_pts, n_lev, COLUMNS)
        prog = native.level_program(stages, dev)
        if strip:
            prog.host_prog, prog.vec_prog = None, {}
        native.pointwise_stack(y.data, y.data, prog=prog, **kw)
        assert torch.equal(y.data.view(torch.uint8), want.data.view(torch.uint8)), (selected, strip)


@pytest.mark.parametrize("layout", LAYOUTS)
def test_pointwise_mask_and_two_stages(dev, layout):
    """R: apply_mask.py:185 after R: rescale.py:25 in one pass; unselected levels untouched."""
    rng = np.random.default_rng(12)
    n_lev, n_pts = 5, 1237
    x = make_fields(rng, n_lev, n_pts, np.float64, nan_frac=0.01)
    mask = rng.random(n_pts) < 0.3
    s0 = [(native.OP_AFFINE, 0, 2.0, 1.0) if l in (1, 2) else (native.OP_COPY, 0, 0, 0) for l in range(n_lev)]
    s1 = [(native.OP_COPY, 1, 0, 0) if l in (2, 4) else (native.OP_COPY, 0, 0, 0) for l in range(n_lev)]
    got = _run_prog(dev, x, layout, [s0, s1], mask=mask)
    want = x.copy()
    want[1] = x[1] * 2.0 + 1.0
    want[2] = oracle.apply_mask_values(x[2] * 2.0 + 1.0, mask)
    want[4] = oracle.apply_mask_values(x[4], mask)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


# ---------------------------------------------------------------------------------
# masks, compaction, reductions
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("op,cmp", [(">", native.CMP_GT), ("<", native.CMP_LT), ("==", native.CMP_EQ),
                                    ("!=", native.CMP_NE), (">=", native.CMP_GE), ("<=", native.CMP_LE)])
def test_mask_build_operators(dev, tdtype, np_dtype, op, cmp):
    """R: apply_mask.py:23-36,160-163 incl. NaN semantics (== false, != true)."""
    rng = np.random.default_rng(13)
    n = 10007
    m = rng.choice(np.array([0.0, 0.25, 0.5, 0.75, 1.0, np.nan]), size=n).astype(np_dtype)
    mask = torch.empty(n + 3, dtype=torch.uint8, device=dev)
    native.mask_build(to_dev(m, dev), mask, n=n, cmp=cmp, threshold=0.5)
    want = oracle.compute_mask(m, threshold=0.5, threshold_operator=op)
    assert np.array_equal(mask[:n].cpu().numpy().astype(bool), want)


def test_mask_build_from_stack_level(dev):
    """A mask field living inside a columns stack is read with stride = pitch."""
    rng = np.random.default_rng(14)
    x = make_fields(rng, 4, 999, np.float64, nan_frac=0.2)
    st = Stack.from_fields(x, dev=dev)
    mask = torch.empty(1000, dtype=torch.uint8, device=dev)
    native.mask_build(st.level_view(2), mask, n=999, stride=st.pitch, cmp=native.CMP_NOTNAN)
    assert np.array_equal(mask[:999].cpu().numpy().astype(bool), oracle.not_nan_mask(x[2]))
    assert native.mask_count(mask, 999) == int(oracle.not_nan_mask(x[2]).sum())


@pytest.mark.parametrize("n", [0, 1, 15, 4096, 4097, 1_000_003, 16_777_216, 16_777_217 + 4096])  # the last two: around the 4096-workgroup limit of the self-scanning scatter
@pytest.mark.parametrize("density", [0.0, 0.37, 1.0])
def test_mask_to_index_is_stable_compaction(dev, n, density):
    """R: remove_nans.py:113 / regrid.py:420 — boolean indexing == gather by ascending index list."""
    rng = np.random.default_rng(15)
    m = (rng.random(n) < density) if 0 < density < 1 else np.full(n, bool(density))
    mask = to_dev(m.astype(np.uint8), dev) if n else torch.empty(0, dtype=torch.uint8, device=dev)
    if n == 0:
        mask = torch.empty(4, dtype=torch.uint8, device=dev)
    index = native.mask_to_index(mask, n)
    assert np.array_equal(index.cpu().numpy(), np.flatnonzero(m).astype(np.int32))


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
def test_reduce(dev, tdtype, np_dtype):
    rng = np.random.default_rng(16)
    x = make_fields(rng, 1, 100003, np_dtype)[0]
    xd = to_dev(x, dev)
    assert native.reduce(xd, native.RED_MIN) == float(x.min())
    assert native.reduce(xd, native.RED_MAX) == float(x.max())
    assert native.reduce(xd, native.RED_NANCOUNT) == 0.0
    x[[5, 77, 99999]] = np.nan
    xd = to_dev(x, dev)
    assert np.isnan(native.reduce(xd, native.RED_MIN)) and np.isnan(x.min())
    assert np.isnan(native.reduce(xd, native.RED_MAX))
    assert native.reduce(xd, native.RED_NANCOUNT) == float(np.sum(np.isnan(x)))
    assert all(np.isnan(v) for v in native.reduce(xd, native.RED_MINMAX))
    # both extremes from ONE pass (the range check of cos_sin_from_rad): flat arrays of awkward lengths and alignments
    # (an unaligned base takes the 4-byte kernels, twice), and pitched stacks in both layouts with poisoned padding
    for n, shift in ((1, 0), (3, 0), (4, 0), (5, 1), (1027, 0), (100003, 3), (100000, 0)):
        y = make_fields(rng, 1, n + shift, np_dtype)[0]
        yd = to_dev(y, dev)[shift:]
        assert native.reduce(yd, native.RED_MINMAX) == (float(y[shift:].min()), float(y[shift:].max())), (n, shift)
        assert native.reduce(yd, native.RED_MIN) == float(y[shift:].min()) and native.reduce(yd, native.RED_NANCOUNT) == 0.0
    # the <= 3 elements after the last whole vector of a flat array go through the scalar kernel (nothing is read past x[n)): the
    # extremes and a NaN placed exactly there must be seen
    for n in (5, 7, 1027, 100003):
        y = make_fields(rng, 1, n, np_dtype)[0]
        y[-1], y[-2] = 1e6, -1e6
        assert native.reduce(to_dev(y, dev), native.RED_MINMAX) == (-1e6, 1e6), n
        y[-1] = np.nan
        yd = to_dev(y, dev)
        assert all(np.isnan(v) for v in native.reduce(yd, native.RED_MINMAX)) and native.reduce(yd, native.RED_NANCOUNT) == 1.0, n
    for layout in LAYOUTS:
        for n_lev, n_pts, pad in ((1, 50, 0), (3, 1001, 1), (137, 997, 3), (13, 4099, 4), (140, 513, 0)):
            z = make_fields(rng, n_lev, n_pts, np_dtype)
            rows, row_len = (n_pts, n_lev) if layout == COLUMNS else (n_lev, n_pts)
            per16 = 16 // z.itemsize
            for pitch in (row_len + pad, (row_len + pad + per16 - 1) // per16 * per16):
                data = torch.full((rows, pitch), float("nan"), dtype=tdtype, device=dev)
                data[:, :row_len] = to_dev(z.T if layout == COLUMNS else z, dev)
                kw = dict(n_pts=n_pts, n_lev=n_lev, pitch=pitch, layout=layout)
                assert native.reduce_stack(data, native.RED_MINMAX, **kw) == (float(z.min()), float(z.max())), (layout, n_lev, n_pts, pitch)
                assert native.reduce_stack(data, native.RED_MAX, **kw) == float(z.max())
                assert native.reduce_stack(data, native.RED_NANCOUNT, **kw) == 0.0


# ---------------------------------------------------------------------------------
# layout
# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("n_lev,n_pts", [(1, 1), (3, 70), (137, 1000), (200, 333), (5, 40320)])
def test_relayout_round_trip(dev, tdtype, np_dtype, n_lev, n_pts):
    rng = np.random.default_rng(17)
    x = make_fields(rng, n_lev, n_pts, np_dtype, nan_frac=0.01)
    fm = Stack.from_fields(x, dev=dev, layout=FIELDS)
    cols = fm.to_layout(COLUMNS)
    itype = np.uint32 if np_dtype == np.float32 else np.uint64
    assert np.array_equal(cols.data[:, :n_lev].T.cpu().numpy().view(itype), x.view(itype))
    back = cols.to_layout(FIELDS)
    assert np.array_equal(back.data.cpu().numpy().view(itype), x.view(itype))
    assert np.array_equal(cols.level_numpy(n_lev - 1).view(itype), x[-1].view(itype))


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("k,n_stack", [(1, 3), (4, 1), (4, 5), (4, 19)])
def test_batched_regrid_equals_one_launch_per_stack(dev, tdtype, np_dtype, layout, k, n_stack):
    """atx_regrid_ell_batch: several stacks through one launch (more than 16 are split internally), bit for bit the
    single-stack results; mismatched shapes fall back to single launches in GatherPlan.apply_many."""
    from anemoi_transform_amd.gather import GatherPlan

    rng = np.random.default_rng(41)
    n_src, n_tgt, n_lev = 4000, 2777, 13
    idx = rng.integers(0, n_src, (n_tgt, k))
    w = None if k == 1 else rng.random((n_tgt, k))
    plan = GatherPlan(n_src, n_tgt, index=idx, weights=w)
    stacks = [Stack.from_fields(make_fields(rng, n_lev, n_src, np_dtype, nan_frac=0.01), dev=dev, layout=layout) for _ in range(n_stack)]
    many = plan.apply_many(stacks)
    itype = torch.int32 if np_dtype == np.float32 else torch.int64
    for st, got in zip(stacks, many):
        want = plan.apply(st)
        assert got.layout == layout and got.n_pts == n_tgt
        cut = (slice(None), slice(0, n_lev)) if layout == COLUMNS else (slice(None), slice(0, n_tgt))
        assert torch.equal(got.data[cut].contiguous().view(itype), want.data[cut].contiguous().view(itype))
    odd = Stack.from_fields(make_fields(rng, n_lev + 2, n_src, np_dtype), dev=dev, layout=layout)
    mixed = plan.apply_many([stacks[0], odd])
    assert mixed[1].n_lev == n_lev + 2 and np.array_equal(mixed[1].numpy(), plan.apply(odd).numpy(), equal_nan=True)
    with pytest.raises(ValueError):
        native.regrid_ell_batch([stacks[0].data], [many[0].data], plan._tensors(dev, tdtype)[0], plan._tensors(dev, tdtype)[1],
                                n_src=n_src, n_tgt=n_tgt, k=65, n_lev=n_lev, src_pitch=stacks[0].pitch, out_pitch=many[0].pitch, layout=layout)


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("n_lev,n_pts,n_out", [(1, 1, 1), (7, 1000, 3), (137, 5003, 137), (300, 700, 290)])
def test_select_levels_and_stack_reductions(dev, tdtype, np_dtype, layout, n_lev, n_pts, n_out):
    """Level gather (re-listing fields of a stack) is a bit copy; untouched levels keep their content;
    reductions over a pitched stack ignore the padding."""
    rng = np.random.default_rng(23)
    x = make_fields(rng, n_lev, n_pts, np_dtype, nan_frac=0.01)
    st = Stack.from_fields(x, dev=dev, layout=layout)
    itype = np.uint32 if np_dtype == np.float32 else np.uint64
    level_map = rng.integers(0, n_lev, n_out).tolist()  # repeats allowed
    if n_out > 2:
        level_map[1] = -1
    out = Stack.empty(n_pts, n_out, tdtype, dev, layout)
    out.data.fill_(7.0)
    native.select_levels(st.data, out.data, level_map, n_pts=n_pts, n_src_lev=n_lev, src_pitch=st.pitch, dst_pitch=out.pitch, layout=layout)
    got = out.numpy()
    want = np.stack([x[l] if l >= 0 else np.full(n_pts, 7.0, np_dtype) for l in level_map])
    assert np.array_equal(got.view(itype), want.view(itype))
    with pytest.raises(ValueError):
        native.select_levels(st.data, out.data, [n_lev] * n_out, n_pts=n_pts, n_src_lev=n_lev, src_pitch=st.pitch, dst_pitch=out.pitch, layout=layout)
    # pitched reductions: poison the padding of a columns stack, it must not be seen
    if layout == COLUMNS and st.pitch > n_lev:
        st.data[:, n_lev:] = float("nan")
    kw = dict(n_pts=n_pts, n_lev=n_lev, pitch=st.pitch, layout=layout)
    assert native.reduce_stack(st.data, native.RED_NANCOUNT, **kw) == float(np.isnan(x).sum())
    clean = np.nan_to_num(x, nan=0.5)
    sc = Stack.from_fields(clean, dev=dev, layout=layout)
    if layout == COLUMNS and sc.pitch > n_lev:
        sc.data[:, n_lev:] = 1e30
    assert native.reduce_stack(sc.data, native.RED_MIN, **kw) == float(clean.min())
    if layout == COLUMNS and sc.pitch > n_lev:
        sc.data[:, n_lev:] = 1e30
    assert native.reduce_stack(sc.data, native.RED_MAX, **kw) == float(clean.max())


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("layout", LAYOUTS)
def test_padded_ragged_rows_run_on_the_fixed_k_kernel(dev, tdtype, np_dtype, layout):
    """Short ragged CSR rows (0..4 entries, like MIR's matrices) become fixed-k rows padded with index -1;
    absent entries are skipped, not multiplied by zero: inf / NaN in the source behave as in scipy."""
    from anemoi_transform_amd.gather import GatherPlan

    rng = np.random.default_rng(31)
    n_src, n_tgt, n_lev = 3000, 2111, 9
    x = make_fields(rng, n_lev, n_src, np_dtype, nan_frac=0.02)
    x[:, rng.integers(0, n_src, 40)] = np.inf
    x[:, rng.integers(0, n_src, 40)] = -np.inf
    lengths = rng.choice([0, 2, 3, 4, 4, 4], size=n_tgt)
    indptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    indices = rng.integers(0, n_src, size=int(indptr[-1])).astype(np.int32)
    data = rng.random(int(indptr[-1]))
    plan = GatherPlan.from_matrix(dict(matrix_data=data, matrix_indices=indices, matrix_indptr=indptr, matrix_shape=(n_tgt, n_src)))
    assert plan.kind == "ell" and plan.padded and plan.k == 4 and (plan.index == -1).any()
    src = Stack.from_fields(x, dev=dev, layout=layout)
    got = plan.apply(src).numpy()
    with np.errstate(invalid="ignore"):
        want = np.stack([oracle.csr_apply(data.astype(np_dtype), indices, indptr, (n_tgt, n_src), f) for f in x])
    assert_interp(got, want, np_dtype)  # padded rows skip their -1 entries: scipy's order and bits in both widths
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isinf(got), np.isinf(want))
    assert (got[:, lengths == 0] == 0).all()  # an empty row is 0, as in scipy
    # shards of a padded plan stay padded and concatenate to the same result
    parts = [plan.shard(r, 3).apply(src).numpy() for r in range(3)]
    assert np.array_equal(np.concatenate(parts, axis=1), got, equal_nan=True)
    # long or very sparse rows keep the general CSR kernel
    long_rows = GatherPlan.from_matrix(dict(matrix_data=np.ones(39), matrix_indices=np.arange(39, dtype=np.int32),
                                            matrix_indptr=np.array([0, 20, 39], dtype=np.int32), matrix_shape=(2, 100)))
    assert long_rows.kind == "csr"


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
@pytest.mark.parametrize("layout", LAYOUTS)
def test_gather_into_a_kept_stack_and_bound_launch(dev, tdtype, np_dtype, layout):
    """`GatherPlan.apply(out=)` and `GatherPlan.bind` (native.BoundCall: arguments converted once, the stream captured) launch the
    same kernel as `apply` — same bits — and a bound launch sees new contents of the source buffer."""
    rng = np.random.default_rng(21)
    n_src, n_tgt, n_lev, k = 4000, 2500, 3, 4
    x = make_fields(rng, n_lev, n_src, np_dtype)
    idx, w = random_ell(rng, n_src, n_tgt, k, np_dtype)
    plan = GatherPlan(n_src, n_tgt, index=idx, weights=w.astype(np.float64))
    src = Stack.from_fields(x, dev=dev, layout=layout)
    indptr = np.arange(n_tgt + 1) * k
    want = np.stack([oracle.csr_apply(w.reshape(-1), idx.reshape(-1), indptr, (n_tgt, n_src), f) for f in x])
    out = src.new_like(n_pts=n_tgt)
    out.data.fill_(float("nan"))
    assert plan.apply(src, out=out) is out
    assert_interp(out.numpy(), want, np_dtype)
    launch, kept = plan.bind(src)
    assert isinstance(launch, native.BoundCall)
    launch()
    assert torch.equal(kept.data, out.data)
    src.data.mul_(2.0)  # exact in binary: every product doubles, so does every sum
    launch()
    torch.cuda.synchronize()
    assert_interp(kept.numpy(), (2.0 * want).astype(np_dtype), np_dtype)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):  # a call bound under another stream enqueues there
        launch2, kept2 = plan.bind(src)
        launch2()
    side.synchronize()
    assert torch.equal(kept2.data, kept.data)


@pytest.mark.parametrize("tdtype,np_dtype", DTYPES)
def test_tall_stacks_with_long_programs_are_served(dev, tdtype, np_dtype):
    """1000 levels x 8 stages of per-level operators: the per-level tables (72 KB in float32, 136 KB in float64) exceed what a
    workgroup may stage in LDS.  Round 3 answered ATX_ENOTIMPL; the entry points now run the stages in halves (per-point programs)
    or gather first and apply the program to the output in place (fused regrid epilogue) — the same statements in the same order,
    so the oracle's bits."""
    import native_double

    rng = np.random.default_rng(33)
    n_lev, n_stage, n_src, n_tgt, k = 1000, 8, 900, 400, 4
    x = make_fields(rng, n_lev, n_src, np_dtype)
    ops = (native.OP_AFFINE, native.OP_MUL, native.OP_CLIP, native.OP_COPY, native.OP_AFFINE_INV)
    stages = []
    for s in range(n_stage):
        stage = []
        for l in range(n_lev):
            op = ops[int(rng.integers(len(ops)))]
            p0, p1 = (float(rng.uniform(200, 280)), float(rng.uniform(281, 360))) if op == native.OP_CLIP else (float(rng.uniform(0.9, 1.1)), float(rng.uniform(-3, 3)))
            stage.append((op, int(rng.random() < 0.2), p0, p1))
        stages.append(stage)
    table = native.LEVEL_OP_DTYPE

    def oracle_program(fields, mask):
        want = np.array(fields, copy=True)
        for stage in stages:
            for l, (op, use_mask, p0, p1) in enumerate(stage):
                entry = np.zeros((), dtype=table)
                entry["op"], entry["use_mask"], entry["p0"], entry["p1"] = op, use_mask, p0, p1
                want[l] = native_double._apply_op(entry, want[l], mask if use_mask else None)
        return want

    prog = native.level_program(stages, dev)
    # per-point program, out of place and in place, column and field-major stacks
    pmask = rng.random(n_src) < 0.3
    pmask_d = to_dev(np.concatenate([pmask, np.zeros(8, bool)]).astype(np.uint8), dev)
    want = oracle_program(x, pmask)
    for layout in LAYOUTS:
        src = Stack.from_fields(x, dev=dev, layout=layout)
        for in_place in (False, True):
            out = src if in_place else src.new_like()
            native.pointwise_stack(src.data, out.data, n_pts=n_src, n_lev=n_lev, x_pitch=src.pitch, y_pitch=out.pitch, layout=layout, prog=prog,
                                   n_stage=n_stage, point_mask=pmask_d)
            assert np.array_equal(out.numpy(), want, equal_nan=True), (layout, in_place)
    # fused regrid epilogue: fixed-k and general CSR
    idx, w = random_ell(rng, n_src, n_tgt, k, np_dtype)
    tmask = rng.random(n_tgt) < 0.3
    tmask_d = to_dev(np.concatenate([tmask, np.zeros(8, bool)]).astype(np.uint8), dev)
    indptr = np.arange(n_tgt + 1) * k
    gathered = np.stack([oracle.csr_apply(w.reshape(-1), idx.reshape(-1), indptr, (n_tgt, n_src), f) for f in x])
    want = oracle_program(gathered, tmask)
    for layout in LAYOUTS:
        src = Stack.from_fields(x, dev=dev, layout=layout)
        out = src.new_like(n_pts=n_tgt)
        native.regrid_ell(src.data, out.data, to_dev(idx, dev), to_dev(w, dev), n_src=n_src, n_tgt=n_tgt, k=k, n_lev=n_lev, src_pitch=src.pitch,
                          out_pitch=out.pitch, layout=layout, prog=prog, n_stage=n_stage, tgt_mask=tmask_d)
        assert np.array_equal(out.numpy(), want, equal_nan=True), ("ell", layout)
        out.data.fill_(0.0)
        native.regrid_csr(src.data, out.data, to_dev(indptr.astype(np.int32), dev), to_dev(idx.reshape(-1), dev), to_dev(w.reshape(-1), dev),
                          n_src=n_src, n_tgt=n_tgt, nnz=n_tgt * k, n_lev=n_lev, src_pitch=src.pitch, out_pitch=out.pitch, layout=layout,
                          prog=prog, n_stage=n_stage, tgt_mask=tmask_d)
        assert np.array_equal(out.numpy(), want, equal_nan=True), ("csr", layout)


def test_validation_mode_refuses_a_table_that_points_outside_the_stack(dev):
    """ATX_VALIDATE=1 (a fresh process: the variable is read once): the library range-checks the tables on the device and refuses the
    launch instead of reading out of bounds — for fixed-k, padded, ordered and CSR tables; valid tables run as usual."""
    import subprocess
    import sys
    import textwrap

    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path.insert(0, %r)
        import __graft_entry__ as g
        g.load_package()
        from anemoi_transform_amd import native
        from anemoi_transform_amd.stack import COLUMNS, Stack
        dev = torch.device("cuda", 0)
        rng = np.random.default_rng(0)
        n_src, n_tgt, n_lev, k = 500, 300, 5, 4
        src = Stack.from_fields(rng.standard_normal((n_lev, n_src)), dev=dev)
        out = src.new_like(n_pts=n_tgt)
        out.data.fill_(7.0)
        idx = rng.integers(0, n_src, (n_tgt, k)).astype(np.int32)
        w = torch.from_numpy(rng.random((n_tgt, k))).to(dev)
        kw = dict(n_src=n_src, n_tgt=n_tgt, k=k, n_lev=n_lev, src_pitch=src.pitch, out_pitch=out.pitch, layout=COLUMNS)
        native.regrid_ell(src.data, out.data, torch.from_numpy(idx).to(dev), w, **kw)          # a valid table runs
        assert not bool((out.data[:, :n_lev] == 7.0).any())
        out.data.fill_(7.0)
        refused = 0
        for bad_value, padded in ((n_src, False), (-1, False), (-2, True), (n_src + 10**6, True)):
            bad = idx.copy(); bad[123, 2] = bad_value
            try:
                native.regrid_ell(src.data, out.data, torch.from_numpy(bad).to(dev), w, padded=padded, **kw)
            except ValueError as e:
                assert "ATX_VALIDATE" in str(e) and "idx" in str(e), e
                refused += 1
        padded_ok = idx.copy(); padded_ok[5, 3] = -1
        native.regrid_ell(src.data, out.data, torch.from_numpy(padded_ok).to(dev), w, padded=True, **kw)   # -1 is the padding marker
        out.data.fill_(7.0)
        rows = torch.arange(n_tgt, dtype=torch.int32, device=dev); rows[7] = n_tgt
        try:
            native.regrid_ell(src.data, out.data, torch.from_numpy(idx).to(dev), w, tgt_rows=rows, **kw)
        except ValueError as e:
            assert "tgt_rows" in str(e); refused += 1
        indptr = (np.arange(n_tgt + 1) * k).astype(np.int32)
        bad = idx.reshape(-1).copy(); bad[77] = n_src
        try:
            native.regrid_csr(src.data, out.data, torch.from_numpy(indptr).to(dev), torch.from_numpy(bad).to(dev), w.reshape(-1), n_src=n_src, n_tgt=n_tgt,
                              nnz=n_tgt * k, n_lev=n_lev, src_pitch=src.pitch, out_pitch=out.pitch, layout=COLUMNS)
        except ValueError as e:
            assert "indices" in str(e); refused += 1
        torch.cuda.synchronize()
        assert bool((out.data[:, :n_lev] == 7.0).all())   # refused launches wrote nothing
        print("refused", refused)
    """ % ROOT)
    env = dict(os.environ, ATX_VALIDATE="1")
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and "refused 6" in run.stdout, run.stdout + run.stderr[-3000:]
