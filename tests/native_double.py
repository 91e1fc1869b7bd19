"""Test double for ``anemoi_transform_amd.native`` — TEST INFRASTRUCTURE.

The product has exactly one compute path (libatx on HBM tensors).  To exercise the
HOST logic (registry, field plumbing, selection, metadata, grouping into stacks,
sharding) on the GPU-less build box, the CPU tests monkeypatch the tensor-level
wrappers of ``native`` with these functions, which evaluate the same contracts on
CPU tensors with the oracle's numpy / scipy statements.  Nothing in the package
imports this module.
"""

from __future__ import annotations

import numpy as np
import torch

from anemoi_transform_amd import native
from oracle import oracle

COLUMNS, FIELDS = native.COLUMNS, native.FIELDS


def _levels(t: torch.Tensor, n_pts: int, n_lev: int, layout: int) -> np.ndarray:
    """Writable [n_lev, n_pts] numpy view of a stack tensor."""
    a = t.numpy()
    return a[:n_pts, :n_lev].T if layout == COLUMNS else a[:n_lev, :n_pts]


def _parse_prog(prog: torch.Tensor, n_stage: int, n_lev: int) -> np.ndarray:
    return prog.numpy().view(native.LEVEL_OP_DTYPE).reshape(n_stage, n_lev)


def _apply_op(entry, x: np.ndarray, mask: np.ndarray | None) -> np.ndarray:
    op, use_mask = int(entry["op"]), int(entry["use_mask"])
    p0, p1 = x.dtype.type(entry["p0"]), x.dtype.type(entry["p1"])
    if op == native.OP_COPY:
        y = x.copy()
    elif op == native.OP_AFFINE:
        y = oracle.rescale_forward(x, p0, p1)
    elif op == native.OP_AFFINE_INV:
        y = oracle.rescale_backward(x, p0, p1)
    elif op == native.OP_MUL:
        y = x * p0
    elif op == native.OP_DIV:
        y = x / p0
    elif op == native.OP_CLIP:
        y = oracle.clip(x, None if np.isnan(p0) else p0, None if np.isnan(p1) else p1)
    elif op == native.OP_IMPUTE_NAN:
        y = oracle.impute_nans(x, p0)
    elif op == native.OP_EXP:
        y = oracle.lnsp_to_sp(x)
    elif op == native.OP_LOG:
        y = oracle.sp_to_lnsp(x)
    elif op == native.OP_SET_NAN:
        y = np.full_like(x, np.nan)
    else:
        raise ValueError(op)
    if use_mask:
        assert mask is not None
        y = oracle.apply_mask_values(y, mask)
    return y


def _epilogue(levels: np.ndarray, prog, n_stage, mask, n_lev):
    if prog is None:
        return
    table = _parse_prog(prog, n_stage, n_lev)
    m = None if mask is None else mask.numpy()[: levels.shape[1]].astype(bool)
    for s in range(n_stage):
        for l in range(n_lev):
            levels[l] = _apply_op(table[s, l], levels[l].copy(), m)


def regrid_ell(src, out, idx, w, *, n_src, n_tgt, k, n_lev, src_pitch, out_pitch, layout, prog=None, n_stage=0, tgt_mask=None,
               padded=False, tgt_rows=None):
    x = _levels(src, n_src, n_lev, layout)
    y = _levels(out, n_tgt, n_lev, layout)
    index = idx.numpy().reshape(n_tgt, k)
    weights = None if w is None else w.numpy().reshape(n_tgt, k)
    if tgt_rows is not None:  # ordered traversal: table row t belongs to output row tgt_rows[t] — undo the permutation of the tables
        rows = tgt_rows.numpy()[:n_tgt]
        assert sorted(rows.tolist()) == list(range(n_tgt)), "tgt_rows must be a permutation"
        back = np.empty(n_tgt, dtype=np.int64)
        back[rows] = np.arange(n_tgt)
        index = index[back]
        weights = None if weights is None else weights[back]
    if weights is None:
        assert k == 1
        for l in range(n_lev):
            y[l] = oracle.gather_nn(x[l], index[:, 0])
    else:
        assert padded or (index >= 0).all()
        present = index >= 0  # ATX_ELL_PADDED: negative index = absent entry of a padded row (atx.h)
        indptr = np.concatenate([[0], np.cumsum(present.sum(axis=1))])
        for l in range(n_lev):
            y[l] = oracle.csr_apply(weights[present], index[present], indptr, (n_tgt, n_src), np.ascontiguousarray(x[l]))
    _epilogue(y, prog, n_stage, tgt_mask, n_lev)


def bind_regrid_ell(src, out, idx, w, *, stream=None, **kw):
    """``native.BoundCall`` stand-in: the same launch, repeated on the same buffers."""
    return lambda: regrid_ell(src, out, idx, w, **kw)


def regrid_ell_batch(srcs, outs, idx, w, **kw):
    for src, out in zip(srcs, outs):
        regrid_ell(src, out, idx, w, **kw)


def regrid_csr(src, out, indptr, indices, data, *, n_src, n_tgt, nnz, n_lev, src_pitch, out_pitch, layout, prog=None,
               n_stage=0, tgt_mask=None, tgt_rows=None):
    x = _levels(src, n_src, n_lev, layout)
    y = _levels(out, n_tgt, n_lev, layout)
    rows = None if tgt_rows is None else tgt_rows.numpy()[:n_tgt]
    for l in range(n_lev):
        r = oracle.csr_apply(data.numpy(), indices.numpy(), indptr.numpy(), (n_tgt, n_src), np.ascontiguousarray(x[l]))
        if rows is None:
            y[l] = r
        else:  # ordered traversal: CSR row t is output row tgt_rows[t]
            y[l][rows] = r
    _epilogue(y, prog, n_stage, tgt_mask, n_lev)


def check_indices(idx, n_src):
    a = idx.numpy()
    return int(((a < 0) | (a >= n_src)).sum())


def pointwise_stack(x, y, *, n_pts, n_lev, x_pitch, y_pitch, layout, prog, n_stage, point_mask=None):
    xs = _levels(x, n_pts, n_lev, layout)
    ys = _levels(y, n_pts, n_lev, layout)
    ys[...] = xs
    _epilogue(ys, prog, n_stage, point_mask, n_lev)


_ORAS6_NAME = {native.ORAS6_KEEP: "siconc", native.ORAS6_ZERO: "siue", native.ORAS6_TEMPERATURE: "sitemptop", native.ORAS6_HEAT: "sihc",
               native.ORAS6_SURFACE: "tos"}


def _oras6_level(x, siconc, kind):
    """One level through ``oracle.oras6_clipping``: the level plays the field of its kind, the other 12 are dummies (the statements of
    the 14 fields do not depend on each other, only on siconc)."""
    if kind == native.ORAS6_CELSIUS:  # the filter decided on the shift; here it is the statement of oras6_clipping.py:191
        x, kind = x + x.dtype.type(oracle.ORAS6_TF), native.ORAS6_TEMPERATURE
    arrays = {name: np.zeros_like(x) for name in oracle.ORAS6_FIELDS}
    arrays["sntemp"] = np.full_like(x, 200.0)  # not in Celsius
    arrays["siconc"] = siconc
    name = _ORAS6_NAME[kind]
    if name != "siconc":
        arrays[name] = x
    return dict(oracle.oras6_clipping(**arrays))[name] if name != "siconc" else x.copy()


def combine_stack(op, inputs, outputs, *, n_pts, n_lev, pitch, layout, level_param=None, flags=0):
    if op == native.COMB_ORAS6:  # the second operand is ONE field shared by the levels
        xs = np.ascontiguousarray(_levels(inputs[0], n_pts, n_lev, layout))
        ys = _levels(outputs[0], n_pts, n_lev, layout)
        siconc = inputs[1].numpy().reshape(-1)[:n_pts]
        for l in range(n_lev):
            ys[l] = _oras6_level(xs[l], siconc.copy(), int(level_param[l]))
        return
    if op == native.COMB_LOOKUP:
        xs = _levels(inputs[0], n_pts, n_lev, layout)
        ys = _levels(outputs[0], n_pts, n_lev, layout)
        n = int(level_param[0])
        table = {c: (float(level_param[1 + c]),) for c in range(n)}
        for l in range(n_lev):
            try:
                ys[l] = oracle.crosswalk(xs[l], table)[0]
            except KeyError:  # the kernel's contract: NaN where the class is not a key
                ys[l] = [table[c][0] if c in table else np.nan for c in xs[l]]
        return
    xs = [np.ascontiguousarray(_levels(t, n_pts, n_lev, layout)) for t in inputs]
    ys = [_levels(t, n_pts, n_lev, layout) for t in outputs]
    deg = bool(flags & native.COMB_DEGREES)
    dt = xs[0].dtype.type
    for l in range(n_lev):
        a = [x[l] for x in xs]
        if op == native.COMB_SNOW_DEPTH_M:
            ys[0][l] = oracle.snow_depth_m(a[0], a[1])
        elif op == native.COMB_SNOW_COVER:
            ys[0][l] = oracle.snow_cover(a[0], a[1])
        elif op == native.COMB_COS_SIN:
            ys[0][l], ys[1][l] = oracle.cos_sin(a[0], deg)
        elif op == native.COMB_ATAN2:
            ys[0][l] = oracle.direction_from_cos_sin(a[0], a[1], deg)
        elif op == native.COMB_W_TO_WZ:
            ys[0][l] = oracle.w_to_wz(a[0], a[1], a[2], float(level_param[l]))
        elif op == native.COMB_WZ_TO_W:
            ys[0][l] = oracle.wz_to_w(a[0], a[1], a[2], float(level_param[l]))
        elif op == native.COMB_SUM:
            ys[0][l] = oracle.sum_fields(a)
        elif op == native.COMB_SUB:
            ys[0][l] = oracle.interval_difference(a[0], a[1])
        elif op == native.COMB_XY_TO_POLAR:
            ys[0][l], ys[1][l] = oracle.xy_to_polar(a[0], a[1])
        elif op == native.COMB_POLAR_TO_XY:
            ys[0][l], ys[1][l] = oracle.polar_to_xy(a[0], a[1])
        elif op == native.COMB_R_TO_D:
            ys[0][l] = oracle.dewpoint_from_relative_humidity(a[0], a[1])
        elif op == native.COMB_D_TO_R:
            ys[0][l] = oracle.relative_humidity_from_dewpoint(a[0], a[1])
        elif op in (native.COMB_Q_TO_R, native.COMB_R_TO_Q):
            pressure = a[2] if len(a) > 2 else dt(100.0) * dt(float(level_param[l]))
            fn = oracle.relative_humidity_from_specific_humidity if op == native.COMB_Q_TO_R else oracle.specific_humidity_from_relative_humidity
            ys[0][l] = fn(a[1], a[0], pressure).astype(dt)
        elif op == native.COMB_OPERA_CLIP:
            ys[0][l], ys[1][l] = oracle.opera_clipping(a[0], a[1], float(level_param[l]))
        elif op == native.COMB_OPERA_PREPROCESS:
            ys[0][l], ys[1][l] = oracle.opera_preprocessing(a[0], a[1], a[2], float(level_param[l]))
        else:
            raise ValueError(op)
    for y in ys:
        assert y.dtype.type == dt


_CMP = {
    native.CMP_GT: ">", native.CMP_LT: "<", native.CMP_EQ: "==", native.CMP_NE: "!=", native.CMP_GE: ">=", native.CMP_LE: "<=",
}


def mask_build(m, mask, *, n, stride=1, cmp, threshold=0.0):
    values = m.numpy().reshape(-1)[: n]  # `m` is already the strided view of one level
    if cmp == native.CMP_NOTNAN:
        res = oracle.not_nan_mask(values)
    elif cmp == native.CMP_ISNAN:
        res = ~oracle.not_nan_mask(values)
    else:
        res = oracle.compute_mask(values, threshold=values.dtype.type(threshold), threshold_operator=_CMP[cmp])
    mask.numpy()[:n] = res.astype(np.uint8)


def mask_count(mask, n=None):
    n = mask.numel() if n is None else n
    return int((mask.numpy()[:n] != 0).sum())


def mask_to_index(mask, n=None):
    n = mask.numel() if n is None else n
    return torch.from_numpy(np.flatnonzero(mask.numpy()[:n]).astype(np.int32))


def reduce(x, red, n=None):
    a = x.numpy().reshape(-1)[: (x.numel() if n is None else n)]
    if red == native.RED_MINMAX:
        return float(a.min()), float(a.max())
    if red == native.RED_MIN:
        return float(a.min())
    if red == native.RED_MAX:
        return float(a.max())
    return float(np.isnan(a).sum())


def relayout(src, dst, *, n_pts, n_lev, src_pitch, dst_pitch, src_layout, dst_layout):
    _levels(dst, n_pts, n_lev, dst_layout)[...] = _levels(src, n_pts, n_lev, src_layout)


def reduce_stack(x, red, *, n_pts, n_lev, pitch, layout):
    a = _levels(x, n_pts, n_lev, layout)
    if red == native.RED_MINMAX:
        return float(a.min()), float(a.max())
    if red == native.RED_MIN:
        return float(a.min())
    if red == native.RED_MAX:
        return float(a.max())
    return float(np.isnan(a).sum())


def select_levels(src, dst, level_map, *, n_pts, n_src_lev, src_pitch, dst_pitch, layout):
    s, d = _levels(src, n_pts, n_src_lev, layout), _levels(dst, n_pts, len(level_map), layout)
    for j, l in enumerate(level_map):
        assert l < n_src_lev
        if l >= 0:
            d[j] = s[l]


class KnnIndex:
    """Brute-force stand-in for the device k-NN index: scipy's squared-distance arithmetic, candidates ordered by
    (distance, source index) — the contract of ``atx_knn_query``."""

    def __init__(self, src_xyz: torch.Tensor) -> None:
        self.src = src_xyz.numpy()
        self.n_src = len(self.src)

    def query(self, tgt_xyz: torch.Tensor, k: int):
        tgt = tgt_xyz.numpy()
        idx = np.full((len(tgt), k), self.n_src, dtype=np.int32)
        d2 = np.full((len(tgt), k), np.inf)
        for t, x in enumerate(tgt):
            diff = self.src - x
            s = np.zeros(self.n_src)
            for c in range(3):
                s = s + diff[:, c] * diff[:, c]
            order = np.lexsort((np.arange(self.n_src), s))[:k]
            idx[t, : len(order)] = order
            d2[t, : len(order)] = s[order]
        return torch.from_numpy(idx), torch.from_numpy(d2)


def cutout_inside(global_xyz, lam_xyz, neighbours):
    g, lam, nb = global_xyz.numpy(), lam_xyz.numpy(), neighbours.numpy()
    n, k = nb.shape
    inside = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        for j in range(k):
            if oracle.triangle_intersect(lam[nb[i, j]], lam[nb[i, (j + 1) % k]], lam[nb[i, (j + 2) % k]], np.zeros(3), g[i]):
                inside[i] = 1
                break
    return torch.from_numpy(inside)


PATCHED = ["KnnIndex", "cutout_inside", "regrid_ell", "bind_regrid_ell", "regrid_ell_batch", "regrid_csr", "check_indices", "pointwise_stack", "combine_stack", "mask_build", "mask_count", "mask_to_index",
           "reduce", "relayout", "reduce_stack", "select_levels"]


def install(monkeypatch) -> None:
    """Route the package's native wrappers and its device choice to this double."""
    from anemoi_transform_amd import stack

    for name in PATCHED:
        monkeypatch.setattr(native, name, globals()[name])
    monkeypatch.setattr(stack, "device", lambda: torch.device("cpu"))
