"""Randomised differential test of the regrid / per-point / level-gather kernels against the oracle.

Seeded, so every run draws the same cases: odd sizes around the vector width, the tile sizes and the workgroup size;
loose pitches (padding poisoned with NaN, which must neither be read into a result nor overwritten); both layouts, both
dtypes; fixed-k, padded ragged and general CSR operators; epilogue programs with point masks; batches.  Float64
results must equal numpy / scipy bit for bit, float32 within the 1e-6 relative tolerance of BASELINE.json.
"""

from __future__ import annotations

import os

import numpy as np
import pytest
import torch

from anemoi_transform_amd import native
from anemoi_transform_amd.stack import COLUMNS, FIELDS
from oracle import oracle

pytestmark = pytest.mark.gpu

POISON = float("nan")
# ATX_RANDOM_SEEDS=first:count widens the sweep for a one-off soak run (default: the seeds of the suite)
_FIRST, _COUNT = (int(v) for v in os.environ.get("ATX_RANDOM_SEEDS", "0:0").split(":"))
REGRID_SEEDS = range(_FIRST, _FIRST + _COUNT) if _COUNT else range(30)
POINTWISE_SEEDS = range(_FIRST, _FIRST + _COUNT) if _COUNT else range(15)


class Loose:
    """A stack with an arbitrary pitch whose padding is poisoned."""

    def __init__(self, values: np.ndarray, layout: int, pad: int, dev, align16: bool):
        n_lev, n_pts = values.shape
        tdt = torch.float32 if values.dtype == np.float32 else torch.float64
        per16 = 16 // values.itemsize
        rows, row_len = (n_pts, n_lev) if layout == COLUMNS else (n_lev, n_pts)
        pitch = row_len + pad
        if align16:
            pitch = (pitch + per16 - 1) // per16 * per16
        self.data = torch.full((rows, pitch), POISON, dtype=tdt, device=dev)
        block = torch.from_numpy(np.ascontiguousarray(values.T if layout == COLUMNS else values)).to(dev)
        self.data[:, :row_len] = block
        self.n_lev, self.n_pts, self.layout, self.pitch, self.row_len = n_lev, n_pts, layout, pitch, row_len

    def values(self) -> np.ndarray:
        a = self.data[:, : self.row_len].cpu().numpy()
        return a.T.copy() if self.layout == COLUMNS else a

    def padding_untouched(self) -> bool:
        # column stacks: the 16-byte vector kernels own the padding levels up to the next vector boundary (atx.h: the
        # content of the padding is unspecified); everything beyond that, and all padding of field-major rows, stays
        start = self.row_len
        if self.layout == COLUMNS:
            per16 = 16 // self.data.element_size()
            start = (self.row_len + per16 - 1) // per16 * per16
        pad = self.data[:, start:]
        return bool(torch.isnan(pad).all().item()) if pad.numel() else True


ENTRIES = [lambda m: (native.OP_COPY, m, 0.0, 0.0), lambda m: (native.OP_AFFINE, m, 1.5, -3.0), lambda m: (native.OP_MUL, m, 9.80665, 0.0),
           lambda m: (native.OP_AFFINE_INV, m, 2.0, 1.0), lambda m: (native.OP_DIV, m, 9.80665, 0.0), lambda m: (native.OP_CLIP, m, 270.0, 300.0),
           lambda m: (native.OP_IMPUTE_NAN, m, -1.0, 0.0)]


def random_program(rng, n_stage, n_lev):
    """Stages in the shapes the kernels tell apart: one operator for all levels; two runs of levels (split anywhere: on and off
    16-byte boundaries); three or four runs with boundaries anywhere (several variables sharing a column: the by-value runs routes of
    round 4); the multiply-add family level by level; a few active levels among COPY; anything goes."""
    ops = []
    for _ in range(n_stage):
        style = rng.choice(["uniform", "two_pieces", "runs", "madd", "sparse", "random"])
        pick = lambda family=7: ENTRIES[int(rng.integers(0, family))](int(rng.random() < 0.3))  # noqa: E731
        if style == "uniform":
            stage = [pick()] * n_lev
        elif style == "two_pieces":
            split = int(rng.integers(0, n_lev + 1))
            if rng.random() < 0.5:
                split = split // 4 * 4
            first, second = pick(), pick()
            stage = [first if l < split else second for l in range(n_lev)]
        elif style == "runs":
            cuts = sorted(int(v) for v in rng.integers(0, n_lev + 1, int(rng.integers(2, 4))))  # 2 or 3 cuts: up to 3 or 4 runs (empty ones allowed)
            run_ops = [pick() for _ in range(len(cuts) + 1)]
            stage = [run_ops[sum(l >= c for c in cuts)] for l in range(n_lev)]
        elif style == "madd":
            stage = [pick(3) for _ in range(n_lev)]
        elif style == "sparse":
            active = set(int(v) for v in rng.integers(0, n_lev, int(rng.integers(1, 4))))
            stage = [pick() if l in active else (native.OP_COPY, 0, 0.0, 0.0) for l in range(n_lev)]
        else:
            stage = [pick() for _ in range(n_lev)]
        if rng.random() < 0.5:  # the same operators with parameters of their own on every level (normalisation per level)
            stage = [(op, m, p0 * (1.0 + 0.03125 * (l % 7)), p1 + 0.5 * (l % 5)) for l, (op, m, p0, p1) in enumerate(stage)]
        ops.append(stage)
    return ops


def apply_program_host(levels: np.ndarray, ops, mask: np.ndarray | None) -> np.ndarray:
    out = levels.copy()
    dt = levels.dtype.type
    for stage in ops:
        for l, (op, use_mask, p0, p1) in enumerate(stage):
            x = out[l]
            p0, p1 = dt(p0), dt(p1)
            if op == native.OP_AFFINE:
                x = oracle.rescale_forward(x, p0, p1)
            elif op == native.OP_AFFINE_INV:
                x = oracle.rescale_backward(x, p0, p1)
            elif op == native.OP_MUL:
                x = x * p0
            elif op == native.OP_DIV:
                x = x / p0
            elif op == native.OP_CLIP:
                x = oracle.clip(x, p0, p1)
            elif op == native.OP_IMPUTE_NAN:
                x = oracle.impute_nans(x.copy(), p0)
            if use_mask and mask is not None:
                x = oracle.apply_mask_values(x.copy(), mask)
            out[l] = x
    return out


def check(got: np.ndarray, want: np.ndarray, what: str):
    if got.dtype == np.float64:
        assert np.array_equal(got, want, equal_nan=True), what
    else:
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-4, err_msg=what)


@pytest.mark.parametrize("seed", REGRID_SEEDS)
def test_random_regrid_cases(dev, seed):
    rng = np.random.default_rng(1000 + seed)
    for case in range(10):
        np_dtype = [np.float32, np.float64][rng.integers(0, 2)]
        layout = [COLUMNS, FIELDS][rng.integers(0, 2)]
        n_lev = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 33, 63, 137, 140, 257]))
        n_src = int(rng.integers(1, 3000))
        n_tgt = int(rng.choice([1, 2, 7, 8, 15, 16, 17, 31, 255, 256, 257, 1000, 2049]))
        pad = int(rng.choice([0, 0, 1, 3, 4, 5]))
        align16 = bool(rng.integers(0, 2))
        kind = rng.choice(["gather", "ell", "padded", "csr"])
        x = (280 + 30 * rng.standard_normal((n_lev, n_src))).astype(np_dtype)
        x[rng.random(x.shape) < 0.02] = np.nan
        src = Loose(x, layout, pad, dev, align16)
        out = Loose(np.zeros((n_lev, n_tgt), np_dtype), layout, int(rng.choice([0, 2, 4])), dev, align16)
        with_prog = rng.random() < 0.4
        ops = random_program(rng, int(rng.integers(1, 4)), n_lev) if with_prog else None
        prog = native.level_program(ops, dev) if with_prog else None
        mask_host = (rng.random(n_tgt) < 0.3) if with_prog else None
        mask_dev = None
        if with_prog:
            mask_dev = torch.zeros(n_tgt + 8, dtype=torch.uint8, device=dev)
            mask_dev[:n_tgt] = torch.from_numpy(mask_host.astype(np.uint8)).to(dev)
        kw = dict(n_src=n_src, n_tgt=n_tgt, n_lev=n_lev, src_pitch=src.pitch, out_pitch=out.pitch, layout=layout,
                  prog=prog, n_stage=len(ops) if with_prog else 0, tgt_mask=mask_dev)
        tile = int(rng.choice([0, 0, 8, 16, 40]))  # 0: built-in choice (direct kernel where it applies); > 0: the tiled kernels
        native.set_tuning(tile)
        what = (f"seed {seed} case {case}: {kind} {np_dtype.__name__} layout {layout} L={n_lev} Ns={n_src} Nt={n_tgt} "
                f"pitches {src.pitch}/{out.pitch} prog={with_prog} tile={tile}")
        if kind == "gather":
            idx = rng.integers(0, n_src, n_tgt).astype(np.int32)
            native.regrid_ell(src.data, out.data, torch.from_numpy(idx).to(dev), None, k=1, **kw)
            want = oracle.gather_nn(x, idx)
        elif kind == "ell":
            k = int(rng.choice([1, 2, 3, 4, 5, 9]))
            idx = rng.integers(0, n_src, (n_tgt, k)).astype(np.int32)
            w = rng.random((n_tgt, k)).astype(np_dtype)
            native.regrid_ell(src.data, out.data, torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev), k=k, **kw)
            indptr = np.arange(n_tgt + 1) * k
            want = np.stack([oracle.csr_apply(w.reshape(-1), idx.reshape(-1), indptr, (n_tgt, n_src), f) for f in x])
        elif kind == "padded":
            k = int(rng.choice([2, 3, 4, 6]))
            idx = rng.integers(0, n_src, (n_tgt, k)).astype(np.int32)
            lengths = rng.integers(0, k + 1, n_tgt)
            present = np.arange(k)[None, :] < lengths[:, None]
            idx[~present] = -1
            w = np.where(present, rng.random((n_tgt, k)), 0.0).astype(np_dtype)
            native.regrid_ell(src.data, out.data, torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev), k=k, padded=True, **kw)
            indptr = np.concatenate([[0], np.cumsum(lengths)])
            want = np.stack([oracle.csr_apply(w[present], idx[present], indptr, (n_tgt, n_src), f) for f in x])
        else:
            lengths = rng.integers(0, int(rng.choice([2, 5, 40])) + 1, n_tgt)
            indptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
            nnz = int(indptr[-1])
            indices = rng.integers(0, n_src, nnz).astype(np.int32)
            data = rng.standard_normal(nnz).astype(np_dtype)
            kw_csr = {k_: v for k_, v in kw.items()}
            native.regrid_csr(src.data, out.data, torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev) if nnz else torch.zeros(1, dtype=torch.int32, device=dev),
                              torch.from_numpy(data).to(dev) if nnz else torch.zeros(1, dtype=src.data.dtype, device=dev), nnz=nnz, **kw_csr)
            want = np.stack([oracle.csr_apply(data, indices, indptr, (n_tgt, n_src), f) for f in x])
        if with_prog:
            want = apply_program_host(want.astype(np_dtype), ops, mask_host)
        native.set_tuning(0)
        check(out.values(), want.astype(np_dtype), what)
        assert out.padding_untouched(), what + " (padding of the output written)"


@pytest.mark.parametrize("seed", POINTWISE_SEEDS)
def test_random_pointwise_and_level_gather_cases(dev, seed):
    rng = np.random.default_rng(2000 + seed)
    for case in range(10):
        np_dtype = [np.float32, np.float64][rng.integers(0, 2)]
        layout = [COLUMNS, FIELDS][rng.integers(0, 2)]
        n_lev = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 31, 137, 300]))
        n_pts = int(rng.choice([1, 3, 4, 5, 63, 64, 65, 255, 257, 1023, 1025, 4099]))
        pad = int(rng.choice([0, 0, 0, 1, 4, 6]))
        align16 = bool(rng.integers(0, 2))
        x = (280 + 30 * rng.standard_normal((n_lev, n_pts))).astype(np_dtype)
        x[rng.random(x.shape) < 0.03] = np.nan
        what = f"seed {seed} case {case}: {np_dtype.__name__} layout {layout} L={n_lev} N={n_pts} pad {pad} align16 {align16}"
        # per-point program, out of place and in place
        ops = random_program(rng, int(rng.integers(1, 5)), n_lev)
        prog = native.level_program(ops, dev)
        mask_host = rng.random(n_pts) < 0.25
        mask_dev = torch.zeros(n_pts + 8, dtype=torch.uint8, device=dev)
        mask_dev[:n_pts] = torch.from_numpy(mask_host.astype(np.uint8)).to(dev)
        src = Loose(x, layout, pad, dev, align16)
        dst = Loose(np.zeros_like(x), layout, pad, dev, align16)
        kw = dict(n_pts=n_pts, n_lev=n_lev, layout=layout, prog=prog, n_stage=len(ops), point_mask=mask_dev)
        native.pointwise_stack(src.data, dst.data, x_pitch=src.pitch, y_pitch=dst.pitch, **kw)
        want = apply_program_host(x, ops, mask_host)
        check(dst.values(), want, what + " out of place")
        assert dst.padding_untouched() and np.array_equal(src.values(), x, equal_nan=True), what
        native.pointwise_stack(src.data, src.data, x_pitch=src.pitch, y_pitch=src.pitch, **kw)
        check(src.values(), want, what + " in place")
        assert src.padding_untouched(), what
        # level gather
        n_out = int(rng.integers(1, n_lev + 3))
        level_map = [int(v) for v in rng.integers(-1, n_lev, n_out)]
        src2 = Loose(x, layout, pad, dev, align16)
        out = Loose(np.full((n_out, n_pts), 7.0, np_dtype), layout, int(rng.choice([0, 3])), dev, align16)
        native.select_levels(src2.data, out.data, level_map, n_pts=n_pts, n_src_lev=n_lev, src_pitch=src2.pitch, dst_pitch=out.pitch, layout=layout)
        want = np.stack([x[l] if l >= 0 else np.full(n_pts, 7.0, np_dtype) for l in level_map])
        itype = np.uint32 if np_dtype == np.float32 else np.uint64
        assert np.array_equal(out.values().view(itype), want.view(itype)), what + " level gather"
        assert out.padding_untouched(), what
        # layout conversion between loose pitches, both directions
        other = FIELDS if layout == COLUMNS else COLUMNS
        conv = Loose(np.zeros_like(x), other, int(rng.choice([0, 1, 4])), dev, align16)
        native.relayout(src2.data, conv.data, n_pts=n_pts, n_lev=n_lev, src_pitch=src2.pitch, dst_pitch=conv.pitch, src_layout=layout, dst_layout=other)
        assert np.array_equal(conv.values().view(itype), x.view(itype)), what + " relayout"
        same = Loose(np.zeros_like(x), layout, int(rng.choice([0, 2])), dev, align16)
        native.relayout(src2.data, same.data, n_pts=n_pts, n_lev=n_lev, src_pitch=src2.pitch, dst_pitch=same.pitch, src_layout=layout, dst_layout=layout)
        assert np.array_equal(same.values().view(itype), x.view(itype)) and same.padding_untouched(), what + " pitched copy"
        # pitched reduction ignores the poisoned padding
        assert native.reduce_stack(src2.data, native.RED_NANCOUNT, n_pts=n_pts, n_lev=n_lev, pitch=src2.pitch, layout=layout) == float(np.isnan(x).sum())


@pytest.mark.parametrize("np_dtype", [np.float32, np.float64])
def test_per_level_programs_on_very_tall_stacks(dev, np_dtype):
    """2001 levels x 2 stages with parameters of their own on every level: the per-level operator tables no longer fit the 64 KB of
    LDS the per-point kernel may use in float64 (68 KB) and the launch falls back to the older chunked kernel — same values either way."""
    rng = np.random.default_rng(5)
    n_lev, n_pts = 2001, 37
    x = (280 + 30 * rng.standard_normal((n_lev, n_pts))).astype(np_dtype)
    ops = [[(native.OP_AFFINE, 0, 1.0 + 0.001 * l, -0.25 * l) for l in range(n_lev)],
           [(native.OP_CLIP, l % 2, 100.0 + 0.1 * l, 300.0) if l % 3 else (native.OP_MUL, 0, 1.0 + 1.0 / (1 + l), 0.0) for l in range(n_lev)]]
    prog = native.level_program(ops, dev)
    mask_host = rng.random(n_pts) < 0.3
    mask_dev = torch.from_numpy(mask_host.astype(np.uint8)).to(dev)
    src, dst = Loose(x, COLUMNS, 0, dev, True), Loose(np.zeros_like(x), COLUMNS, 0, dev, True)
    kw = dict(n_pts=n_pts, n_lev=n_lev, layout=COLUMNS, prog=prog, n_stage=2, point_mask=mask_dev)
    native.pointwise_stack(src.data, dst.data, x_pitch=src.pitch, y_pitch=dst.pitch, **kw)
    want = apply_program_host(x, ops, mask_host)
    check(dst.values(), want, "out of place")
    native.pointwise_stack(src.data, src.data, x_pitch=src.pitch, y_pitch=src.pitch, **kw)
    check(src.values(), want, "in place")


def assert_wrong_side_only_at_the_jump(got, want, sd, rsn):
    """`snow_cover[snow_cover > 0.99] = 1.0` (R: snow_cover.py:38): an element where the device and numpy fall on DIFFERENT sides of the
    jump (one says 1.0, the other the tanh value) is tolerated only where the true tanh of the statement's own argument — evaluated in
    extended precision — lies within 8 eps of 0.99, i.e. where the last bits of two correct tanh routines decide.  A regression that
    moves the threshold, or a tanh off by more than a few ulp near 0.99, fails here however many elements the count guard allows."""
    dt = got.dtype.type
    wrong = (got == 1.0) != (want == 1.0)
    wrong &= ~(np.isnan(got) | np.isnan(want))
    if not wrong.any():
        return
    with np.errstate(all="ignore"):
        arg = (4000 * ((1000 * sd) / rsn)) / np.clip(rsn, 100, 400)  # the statement's argument, in the statement's own width
    exact = np.tanh(arg[wrong].astype(np.longdouble))
    assert np.all(np.abs(exact - np.longdouble(dt(0.99))) <= 8 * np.finfo(dt).eps), (arg[wrong], got[wrong], want[wrong])
    # and on its own side of the jump the device value is a correct tanh
    below = got[wrong] != 1.0
    assert np.all(np.abs(got[wrong][below].astype(np.longdouble) - exact[below]) <= 8 * np.finfo(dt).eps)


@pytest.mark.parametrize("np_dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_snow_cover_around_its_jump(dev, np_dtype):
    """The float32 AND float64 twins of the jump guard, on inputs MADE to sit at the jump: ladders of adjacent snow depths that walk the
    statement's tanh through 0.99 one ulp at a time, for densities below, inside and above the clip of `snow_density` (R: snow_cover.py:36).
    Away from the 8-eps band the device equals numpy to the usual tolerance; inside it every value is one of the statement's own two
    (1.0, or the tanh value itself), and wherever device and numpy disagree about the side, the true tanh is within 8 eps of 0.99."""
    eps = np.finfo(np_dtype).eps
    rsn_values = np.array([60.0, 100.0, 187.5, 250.0, 400.0, 733.0], dtype=np_dtype)
    atanh099 = 0.5 * np.log(1.99 / 0.01)  # 2.6466524123622457
    # offsets in representable numbers from the centre: every one of the nearest 64 on either side (tanh moves by ~0.05 eps per step
    # there: its slope at atanh(0.99) is 1 - 0.99^2), then powers of two out to 2^20 steps — well outside the band on both sides
    far = 2 ** np.arange(7, 21)
    steps = np.concatenate([-far[::-1], np.arange(-64, 65), far]).astype(np.int64)
    bits = np.int32 if np_dtype == np.float32 else np.int64
    sd_rows, rsn_rows = [], []
    for rsn in rsn_values:
        centre = np_dtype(atanh099 * float(rsn) * float(np.clip(rsn, 100, 400)) / 4.0e6)  # 4000 * (1000 * sd / rsn) / clip(rsn) == atanh(0.99)
        ladder = (np.full(steps.size, centre, dtype=np_dtype).view(bits) + steps.astype(bits)).view(np_dtype)  # positive floats: +1 bit = next number
        assert np.all(np.diff(ladder) > 0) and np.nextafter(centre, np_dtype(np.inf)) == ladder[list(steps).index(1)]
        sd_rows.append(ladder)
        rsn_rows.append(np.full(steps.size, rsn, dtype=np_dtype))
    sd, rsn = np.stack(sd_rows), np.stack(rsn_rows)  # [6 levels, 157 points]
    for layout in (COLUMNS, FIELDS):
        a, b = Loose(sd, layout, 0, dev, True), Loose(rsn, layout, 0, dev, True)
        out = Loose(np.zeros_like(sd), layout, 0, dev, True)
        native.combine_stack(native.COMB_SNOW_COVER, [a.data, b.data], [out.data], n_pts=sd.shape[1], n_lev=sd.shape[0], pitch=a.pitch, layout=layout)
        got, want = out.values(), oracle.snow_cover(sd.copy(), rsn.copy())
        with np.errstate(all="ignore"):
            before_jump = np.clip(np.tanh((4000 * ((1000 * sd) / rsn)) / np.clip(rsn, 100, 400)), 0, 1)
        at_jump = np.abs(before_jump - np_dtype(0.99)) <= 8 * eps
        assert at_jump.sum() >= 6 and (~at_jump).sum() >= 6  # the ladders cross the band and leave it on both sides
        assert (want == 1.0).any() and (want < 1.0).any()
        np.testing.assert_allclose(got[~at_jump], want[~at_jump], rtol=1e-13 if np_dtype == np.float64 else 2e-6)
        assert np.all((got[at_jump] == 1.0) | (np.abs(got[at_jump] - before_jump[at_jump]) <= 8 * eps))
        assert np.all((got == 1.0) | (got <= np_dtype(0.99) + 8 * eps))  # nothing strictly between the jump's two values survives
        assert_wrong_side_only_at_the_jump(got, want, sd, rsn)
        # monotone in the snow depth: once a ladder has reached 1.0 it stays there
        for row in got:
            first_one = int(np.argmax(row == 1.0)) if (row == 1.0).any() else row.size
            assert np.all(row[first_one:] == 1.0) and np.all(np.diff(row[:first_one]) >= -8 * eps)


COMBINE_SEEDS = range(_FIRST, _FIRST + _COUNT) if _COUNT else range(12)


@pytest.mark.parametrize("seed", COMBINE_SEEDS)
def test_random_combine_cases(dev, seed):
    """The multi-input per-point kernels (one per operator since round 3) on random shapes, layouts and pitches — 16-byte vector
    and scalar forms, tails, poisoned padding — against the oracle's statements."""
    rng = np.random.default_rng(7000 + seed)
    np_dtype = np.float64 if rng.random() < 0.5 else np.float32
    layout = COLUMNS if rng.random() < 0.6 else FIELDS
    n_lev = int(rng.integers(1, 40))
    n_pts = int(rng.choice([1, 2, 3, 5, 63, 64, 257, 1000, 4099, 20011]))
    pad, align16 = int(rng.integers(0, 6)), bool(rng.random() < 0.6)
    rtol = 1e-13 if np_dtype == np.float64 else 2e-6
    shape = (n_lev, n_pts)
    sd = rng.uniform(0.0, 0.3, shape).astype(np_dtype)
    sd[rng.random(shape) < 0.3] = 0.0  # bare ground: the exact-zero shortcut
    rsn = rng.uniform(50.0, 600.0, shape).astype(np_dtype)
    ang = rng.uniform(-6.3, 6.3, shape).astype(np_dtype)
    t = (250.0 + 40.0 * rng.random(shape)).astype(np_dtype)
    w = rng.normal(0, 0.5, shape).astype(np_dtype)
    q = rng.uniform(0.0, 0.02, shape).astype(np_dtype)
    levels = np.sort(rng.uniform(1.0, 1000.0, n_lev))

    def run(op, ins, n_out, flags=0, with_levels=False):
        stacks = [Loose(x, layout, pad, dev, align16) for x in ins]
        outs = [Loose(np.full(shape, 7.0, dtype=np_dtype), layout, pad, dev, align16) for _ in range(n_out)]
        lp = torch.from_numpy(levels).to(dev) if with_levels else None
        native.combine_stack(op, [s.data for s in stacks], [o.data for o in outs], n_pts=n_pts, n_lev=n_lev, pitch=stacks[0].pitch, layout=layout,
                             level_param=lp, flags=flags)
        for o in outs:  # atx.h: the padding of the outputs is written with zeros (or left alone where no vector reaches it)
            padding = o.data[:, o.row_len:]
            assert bool(((padding == 0) | torch.isnan(padding)).all().item()), (op, layout, pad, align16)
        return [o.values() for o in outs]

    assert np.array_equal(run(native.COMB_SNOW_DEPTH_M, [sd, rsn], 1)[0], oracle.snow_depth_m(sd, rsn))
    assert np.array_equal(run(native.COMB_SUB, [t, w], 1)[0], t - w)
    # snow_cover has a JUMP in its statement — `snow_cover[snow_cover > 0.99] = 1.0` (R: snow_cover.py:38) — so wherever tanh lands within
    # a few ulp of 0.99 the last bit of tanh decides between 0.99 and 1.0: numpy's and the device library's float32 tanh may disagree
    # there (seed 156 of a 300-seed soak: 1 element of 520 286).  Everywhere else the tolerance is the usual one; at such a point both
    # sides of the jump are the statement's own values.
    got, want = run(native.COMB_SNOW_COVER, [sd, rsn], 1)[0], oracle.snow_cover(sd, rsn)
    with np.errstate(all="ignore"):
        before_jump = np.clip(np.tanh((4000 * ((1000 * sd) / rsn)) / np.clip(rsn, 100, 400)), 0, 1)
    at_jump = np.abs(before_jump - np_dtype(0.99)) <= 8 * np.finfo(np_dtype).eps
    np.testing.assert_allclose(got[~at_jump], want[~at_jump], rtol=rtol, atol=1e-7)
    assert np.all((got[at_jump] == 1.0) | (np.abs(got[at_jump] - before_jump[at_jump]) <= 8 * np.finfo(np_dtype).eps))
    # (the exemption must stay an exception: about one element of a 33 000-element float32 case falls into the 8-eps band around 0.99, four
    # did in seed 20482 of round 5's soak — a Poisson tail, every one of them on a side of the jump — so the guard leaves room for that.
    # In float64 the band is 1.8e-15 wide: a random case has no business in it at all)
    assert at_jump.sum() <= (max(8, 3e-4 * at_jump.size) if np_dtype == np.float32 else 1)
    assert_wrong_side_only_at_the_jump(got, want, sd, rsn)
    deg = bool(rng.random() < 0.5)
    x = np.rad2deg(ang).astype(np_dtype) if deg else ang
    co, si = run(native.COMB_COS_SIN, [x], 2, flags=native.COMB_DEGREES if deg else 0)
    wc, ws = oracle.cos_sin(x, deg)
    atol = 1e-6 if np_dtype == np.float32 else 1e-15
    np.testing.assert_allclose(co, wc, rtol=rtol, atol=atol)
    np.testing.assert_allclose(si, ws, rtol=rtol, atol=atol)
    back = run(native.COMB_ATAN2, [wc.astype(np_dtype), ws.astype(np_dtype)], 1, flags=native.COMB_DEGREES if deg else 0)[0]
    np.testing.assert_allclose(back, oracle.direction_from_cos_sin(wc.astype(np_dtype), ws.astype(np_dtype), deg), rtol=rtol, atol=1e-4)
    want = np.stack([oracle.w_to_wz(w[l], t[l], q[l], np_dtype(levels[l])) for l in range(n_lev)])
    np.testing.assert_allclose(run(native.COMB_W_TO_WZ, [w, t, q], 1, with_levels=True)[0], want, rtol=rtol)
    want = np.stack([oracle.wz_to_w(w[l], t[l], q[l], np_dtype(levels[l])) for l in range(n_lev)])
    np.testing.assert_allclose(run(native.COMB_WZ_TO_W, [w, t, q], 1, with_levels=True)[0], want, rtol=rtol)
    n_terms = int(rng.integers(1, 9))
    terms = [sd, rsn, ang, t, w, q, sd * 2, t * 3][:n_terms]
    assert np.array_equal(run(native.COMB_SUM, terms, 1)[0], np.stack([oracle.sum_fields([x[l] for x in terms]) for l in range(n_lev)]))
