"""Gather plans: host-validated index(+weight) tables and their device copies.

One object for the three shapes of the reference's regrid operator
(R: filters/fields/regrid.py):
  * k = 1 index list      — ``data[..., nearest_grid_points]`` (:380), ``data[..., mask]`` (:420),
                            ``data[self._mask]`` of remove_nans (remove_nans.py:113)
  * fixed-k weights (ELL) — a CSR matrix whose rows all hold k entries (:310)
  * general CSR           — any other MIR matrix (:310)

Indices are validated on the host before anything is uploaded (the kernels trust
them): every index must lie in ``[0, n_src)``; cKDTree's "no neighbour" marker
``n_src`` (R: spatial.py:630-632) is rejected here.  Device copies are cached per
(device, dtype).  ``shard(rank, world)`` cuts a contiguous slice of target points
for the multi-GPU path: rows are independent, so no exchange is
needed after the gather.
"""

from __future__ import annotations

from typing import Any

import numpy as np
import torch

from . import native
from .stack import COLUMNS, Stack


# Cost of one target column relative to one first-touched source column in ``GatherPlan.bounds``.  By bytes alone the stored
# column plus its k table entries are 1.06 column-equivalents at 137 f32 levels, but stores stream while gathered reads do not:
# timing every shard of the 4- and 8-way cut on MI355X (O1280 -> 0.25 degree, tools/shard_efficiency.py <world> <cost>,
# batched launch of the direct kernel) the slowest rank is fastest for weights of 0.5-0.7 (0.448-0.451 ms at 8 shards against
# 0.455 at 1.1-1.4 and 0.57 for equal counts).
TARGET_COST = 0.6
# The same weight for a step that is ONE SHORT launch per rank — the strong-scaling split of a single stack (bench.py's headline at
# N > 1: 0.1 ms per rank at 8 ranks): a launch then pays per workgroup as well as per byte, and the polar shards — few source
# columns, twice the targets of the others — are the slow ones under 0.6.  tools/experiments/shard_weight_sweep.py,
# profiles/r05_shard_weight_sweep.log (every shard timed alone, O1280 -> 0.25 degree, 137 levels, float64): slowest of 8 shards 0.126 ms
# at 0.6, 0.119 at 1.1, **0.117 at 1.3**, 0.118 at 1.5-1.8 (speed-up bound of the 8-rank line 6.67 -> 7.22); 4 shards: flat from 0.9 on; float32
# keeps improving up to 1.8 (7.67 -> 8.5).  The batched step prefers 0.6-0.9 on the same tables, so the weight is an argument of
# ``bounds`` / ``shard``, not a new default.
TARGET_COST_SHORT_LAUNCH = 1.3


class GatherPlan:
    def __init__(
        self,
        n_src: int,
        n_tgt: int,
        *,
        index: np.ndarray | None = None,  # [n_tgt] or [n_tgt, k]
        weights: np.ndarray | None = None,  # [n_tgt, k] or None (pure gather)
        csr: tuple[np.ndarray, np.ndarray, np.ndarray] | None = None,  # (data, indices, indptr)
        padded: bool = False,  # index -1 marks an absent entry of a padded ragged row
    ) -> None:
        self.n_src = int(n_src)
        self.n_tgt = int(n_tgt)
        self.padded = bool(padded)
        self.order: np.ndarray | None = None  # visiting order of the targets on the device (order_targets); results never depend on it
        self._device: dict[tuple[str, torch.dtype], tuple[torch.Tensor, ...]] = {}
        if csr is not None:
            data, indices, indptr = csr
            self.kind = "csr"
            self.data = np.ascontiguousarray(data, dtype=np.float64)
            self.indices = self._check(np.ascontiguousarray(indices))
            self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
            if len(self.indptr) != self.n_tgt + 1 or self.indptr[0] != 0 or self.indptr[-1] != len(self.indices):
                raise ValueError("malformed CSR matrix: indptr does not match the shape / number of entries")
            if np.any(np.diff(self.indptr) < 0):
                raise ValueError("malformed CSR matrix: indptr is not non-decreasing")
            if len(self.indices) >= 2**31:
                raise NotImplementedError("matrices with 2^31 or more entries are not supported")
            self.k = None
        else:
            index = np.asarray(index)
            if index.ndim == 1:
                index = index.reshape(-1, 1)
            if index.shape[0] != self.n_tgt:
                raise ValueError(f"index table has {index.shape[0]} rows for {self.n_tgt} target points")
            self.kind = "ell"
            self.k = int(index.shape[1])
            if self.padded:
                if weights is None:
                    raise ValueError("padded rows need weights")
                present = index >= 0
                if np.any(index[~present] != -1):
                    raise ValueError("padded rows use -1 for absent entries")
                self._check(index[present])
                self.index = np.ascontiguousarray(index).astype(np.int32)
            else:
                self.index = self._check(np.ascontiguousarray(index))
            self.weights = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64).reshape(index.shape)
            if self.weights is None and self.k != 1:
                raise ValueError("a gather without weights needs exactly one index per target point")
            if self.weights is not None and self.k in (9, 10, 11, 13, 14, 15):
                # widths between the direct kernel's compile-time forms: absent entries (index -1, skipped in the sum, so the bits are
                # those of the k given) up to 12 or 16 — all k source vectors in flight instead of the tiled kernel's run-time loop
                wide = 12 if self.k < 12 else 16
                fill = np.full((self.n_tgt, wide - self.k), -1, dtype=np.int32)
                self.index = np.ascontiguousarray(np.concatenate([self.index, fill], axis=1))
                self.weights = np.ascontiguousarray(np.concatenate([self.weights, np.zeros(fill.shape)], axis=1))
                self.k, self.padded = wide, True

    # ---- construction helpers ----------------------------------------------------------
    def _check(self, idx: np.ndarray) -> np.ndarray:
        if idx.dtype.kind not in "iu":
            raise ValueError(f"indices must be integers, got {idx.dtype}")
        if idx.size and (idx.min() < 0 or idx.max() >= self.n_src):
            raise ValueError(
                f"gather index outside [0, {self.n_src}): min {idx.min()}, max {idx.max()} "
                "(cKDTree reports len(source) when no neighbour lies within max_distance)"
            )
        if self.n_src >= 2**31:
            raise NotImplementedError("grids with 2^31 or more points are not supported")
        return idx.astype(np.int32)

    @classmethod
    def from_matrix(cls, matrix: dict[str, Any]) -> "GatherPlan":
        """From the npz dict of a regrid matrix (R: regrid.py:281-285)."""
        from .interp import csr_uniform_k

        n_tgt, n_src = (int(s) for s in matrix["matrix_shape"])
        data, indices, indptr = matrix["matrix_data"], matrix["matrix_indices"], matrix["matrix_indptr"]
        indptr = np.asarray(indptr)
        k = csr_uniform_k(indptr)
        if k is not None and k <= 64 and len(indptr) == n_tgt + 1:
            return cls(n_src, n_tgt, index=np.asarray(indices).reshape(n_tgt, k), weights=np.asarray(data).reshape(n_tgt, k))
        padded = cls._padded_rows(n_src, n_tgt, np.asarray(data), np.asarray(indices), indptr)
        if padded is not None:
            return padded
        return cls(n_src, n_tgt, csr=(data, indices, indptr))

    @classmethod
    def _padded_rows(cls, n_src, n_tgt, data, indices, indptr, max_k: int = 16, min_fill: float = 0.6) -> "GatherPlan | None":
        """Ragged rows as fixed-k rows padded with the index -1 (skipped by the kernel): the fast
        fixed-k kernel, the CSR summation order, no arithmetic on the padding.  Widths the direct kernel has compile-time forms for:
        up to 8 as they are, 9-12 padded to 12, 13-16 to 16 (round 3: rows of 9-16 entries 0.50-0.55 on the general CSR kernel)."""
        if len(indptr) != n_tgt + 1 or n_tgt == 0:
            return None
        lengths = np.diff(indptr.astype(np.int64))
        k = int(lengths.max()) if lengths.size else 0
        if 8 < k <= max_k:
            k = 12 if k <= 12 else 16
        if k < 1 or k > max_k or lengths.sum() < min_fill * k * n_tgt:
            return None
        cols = np.arange(k)[None, :]
        present = cols < lengths[:, None]
        pos = np.minimum(indptr[:-1].astype(np.int64)[:, None] + cols, max(len(indices) - 1, 0))
        index = np.where(present, np.asarray(indices, dtype=np.int64)[pos], -1)
        weights = np.where(present, np.asarray(data, dtype=np.float64)[pos], 0.0)
        return cls(n_src, n_tgt, index=index, weights=weights, padded=True)

    @classmethod
    def from_mask(cls, mask: np.ndarray, n_src: int | None = None) -> "GatherPlan":
        """From a boolean mask or an integer index list (R: regrid.py:402,420; spatial.py:533-536)."""
        mask = np.asarray(mask)
        if mask.dtype == bool:
            n_src = mask.size if n_src is None else n_src
            index = np.flatnonzero(mask.reshape(-1))
        else:
            if n_src is None:
                raise ValueError("an integer mask needs the number of source points")
            index = mask.reshape(-1)
            if index.dtype.kind not in "iu":  # numpy's own refusal of `data[..., np.array([1.0, 2.0])]`
                raise IndexError("arrays used as indices must be of integer (or boolean) type")
            if index.size and (int(index.min()) < -n_src or int(index.max()) >= n_src):
                # R: regrid.py:420 `data[..., self.mask]` — numpy's own error for an index list made for another grid
                bad = int(index.max()) if int(index.max()) >= n_src else int(index.min())
                raise IndexError(f"index {bad} is out of bounds for axis 0 with size {n_src}")
            index = np.where(index < 0, index + n_src, index)  # numpy indexing accepts negatives
        return cls(n_src, len(index), index=index)

    def bounds(self, world: int, target_cost: float | None = None) -> list[int]:
        """Target boundaries ``b[0] = 0 <= ... <= b[world] = n_tgt`` of the ``world`` contiguous shards,
        balanced by estimated HBM TRAFFIC rather than by target count.

        A shard's cost is the number of distinct source columns it reads plus (weighted) the target columns
        it writes.  On a lat-lon target grid equal-count shards are badly unbalanced — near the poles many
        targets share few source columns: measured 0.31 ms (polar) vs 0.57 ms (equatorial) per step on
        O1280 -> 0.25 degree at 8 shards, which would cap weak scaling at 82 %; balanced: within a few %
        (profiles/r01_shard_balance.log).  ``target_cost``: the weight of a target column (default ``TARGET_COST``; a step made
        of one short launch per rank balances better with ``TARGET_COST_SHORT_LAUNCH``).
        """
        weight = TARGET_COST if target_cost is None else float(target_cost)
        cached = self.__dict__.setdefault("_bounds", {})
        key = (world, weight)
        if key not in cached:
            if world <= 1 or self.n_tgt == 0:
                cached[key] = [0] + [self.n_tgt] * max(world, 1)
            else:
                if self.kind == "ell":
                    flat = self.index.reshape(-1).astype(np.int64)
                    row_of = np.arange(flat.size) // self.k
                    valid = flat >= 0
                    flat, row_of = flat[valid], row_of[valid]
                else:
                    flat = self.indices.astype(np.int64)
                    row_of = np.repeat(np.arange(self.n_tgt), np.diff(self.indptr))
                # source columns first referenced by each target, in target order
                _, first = np.unique(flat, return_index=True)
                new_sources = np.bincount(row_of[first], minlength=self.n_tgt).astype(np.float64)
                cost = np.cumsum(new_sources + weight)
                edges = np.searchsorted(cost, cost[-1] * np.arange(1, world) / world, side="left") + 1
                b = [0] + [int(min(max(e, 0), self.n_tgt)) for e in edges] + [self.n_tgt]
                for i in range(1, len(b)):
                    b[i] = max(b[i], b[i - 1])
                cached[key] = b
        return cached[key]

    def shard_range(self, rank: int, world: int, target_cost: float | None = None) -> tuple[int, int]:
        b = self.bounds(world, target_cost)
        return b[rank], b[rank + 1]

    def shard(self, rank: int, world: int, target_cost: float | None = None) -> "GatherPlan":
        """The ``rank``-th of ``world`` contiguous, traffic-balanced slices of the target points.  Remembered per (rank, world):
        a job asks for its shard at every step, and a fresh plan would re-validate its indices on the host and upload its tables
        again (O1280 -> 0.25 degree: 4 ms per call against 0.85 ms of kernel)."""
        cached = self.__dict__.setdefault("_shards", {})
        key = (int(rank), int(world), TARGET_COST if target_cost is None else float(target_cost))
        if key not in cached:
            if len(cached) >= 64:
                cached.pop(next(iter(cached)))
            cached[key] = self._cut_shard(rank, world, target_cost)
        return cached[key]

    def _cut_shard(self, rank: int, world: int, target_cost: float | None = None) -> "GatherPlan":
        lo, hi = self.shard_range(rank, world, target_cost)
        if self.kind == "ell":
            part = GatherPlan(self.n_src, hi - lo, index=self.index[lo:hi],
                              weights=None if self.weights is None else self.weights[lo:hi], padded=self.padded)
            if self.order is not None:  # the slice keeps the visiting order of its own targets
                inside = self.order[(self.order >= lo) & (self.order < hi)]
                part.order = (inside - lo).astype(np.int32)
            return part
        p0, p1 = int(self.indptr[lo]), int(self.indptr[hi])
        part = GatherPlan(self.n_src, hi - lo, csr=(self.data[p0:p1], self.indices[p0:p1], self.indptr[lo:hi + 1] - p0))
        if self.order is not None:
            inside = self.order[(self.order >= lo) & (self.order < hi)]
            part.order = (inside - lo).astype(np.int32)
        return part

    # ---- device side -------------------------------------------------------------------
    def order_targets(self, order: np.ndarray | None) -> "GatherPlan":
        """Visit the targets in ``order`` (a permutation of ``range(n_tgt)``; ``None``: natural order) on the device —
        ``atx_regrid_ell_ordered``.  Results are identical; what changes is which targets run together: a lat-lon grid visited in
        column blocks (``column_block_order``) lets vertically adjacent targets, whose neighbour patches overlap, meet in an XCD's
        L2 (O1280 -> 0.25 degree: k = 16 +9-12 %, k = 8 +5-9 %; k <= 4 loses 2-7 % to the scattered output rows, so the library's own
        policy, ``target_order_for``, orders long rows only).  Column stacks only; field-major stacks ignore the order."""
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.int64)
            if order.shape != (self.n_tgt,) or not np.array_equal(np.sort(order), np.arange(self.n_tgt)):
                raise ValueError("order must be a permutation of range(n_tgt)")
            order = order.astype(np.int32)
        self.order = order
        self._device.clear()
        self.__dict__.pop("_shards", None)  # shards carry the visiting order of their own targets
        self.__dict__.pop("_bands", None)
        return self

    def _tensors(self, device: torch.device, dtype: torch.dtype, ordered: bool = False) -> tuple[torch.Tensor, ...]:
        ordered = bool(ordered and self.order is not None)
        key = (str(device), dtype, ordered)
        if key not in self._device:
            np_dtype = np.float32 if dtype == torch.float32 else np.float64
            if self.kind == "ell":
                index, weights = self.index, self.weights
                if ordered:
                    index, weights = index[self.order], None if weights is None else weights[self.order]
                idx = torch.from_numpy(np.ascontiguousarray(index).reshape(-1)).to(device)
                w = None if weights is None else torch.from_numpy(np.ascontiguousarray(weights).astype(np_dtype).reshape(-1)).to(device)
                rows = torch.from_numpy(self.order).to(device) if ordered else None
                self._device[key] = (idx, w, rows)
            else:
                indptr, indices, data = self.indptr, self.indices, self.data
                if ordered:  # the CSR rows in visiting order: row i of the permuted matrix is row order[i] of the original
                    lengths = np.diff(indptr)[self.order]
                    new_ptr = np.concatenate([[0], np.cumsum(lengths)])
                    take = np.repeat(indptr[:-1][self.order] - new_ptr[:-1], lengths) + np.arange(int(new_ptr[-1]))
                    indptr, indices, data = new_ptr, indices[take], data[take]
                self._device[key] = (
                    torch.from_numpy(np.ascontiguousarray(indptr).astype(np.int32)).to(device),
                    torch.from_numpy(np.ascontiguousarray(indices)).to(device),
                    torch.from_numpy(np.ascontiguousarray(data).astype(np_dtype)).to(device),
                    torch.from_numpy(self.order).to(device) if ordered else None,
                )
        return self._device[key]

    def _long_rows(self) -> bool:
        """More than 8 entries per row (on average, for ragged matrices)."""
        if self.kind == "ell":
            return self.k > 8
        return len(self.indices) > 8 * max(self.n_tgt, 1)

    def _check_out(self, src: Stack, out: Stack) -> None:
        if (out.n_pts, out.n_lev, out.dtype, out.layout, out.device) != (self.n_tgt, src.n_lev, src.dtype, src.layout, src.device):
            raise ValueError(f"out= must be a {src.layout_name} stack of {self.n_tgt} points x {src.n_lev} levels, {src.dtype}, on {src.device}; "
                             f"got {out.n_pts} points x {out.n_lev} levels, {out.dtype}, layout {out.layout_name}, on {out.device}")

    def apply(self, src: Stack, *, prog: torch.Tensor | None = None, n_stage: int = 0,
              tgt_mask: torch.Tensor | None = None, out: Stack | None = None) -> Stack:
        """Run the gather over every level of ``src``; returns a new stack on the target points — or ``out``, a stack of the
        right shape the caller keeps across calls (no allocation inside a repeated, launch-bound call)."""
        # R: regrid.py:377-378 — the field must live on the plan's source grid
        assert src.n_pts == self.n_src, (src.n_pts, self.n_src)
        if out is not None:
            self._check_out(src, out)
        if src.layout != COLUMNS and self.n_tgt > 0 and self._long_rows():
            # field-major stacks and rows beyond 8 entries: through column stacks and back — the field-major gather re-fetches a source
            # point once per row that uses it (O1280 -> 0.25 deg, k = 16, 137 fields: 11.5 ms direct, 2.5 ms with both conversions)
            res = self.apply(src.to_layout(COLUMNS), prog=prog, n_stage=n_stage, tgt_mask=tgt_mask).to_layout(src.layout)
            if out is None:
                return res
            out.data.copy_(res.data)
            return out
        if out is None:
            out = src.new_like(n_pts=self.n_tgt, zero=False)
        if self.n_tgt == 0:
            return out
        if self.kind == "ell":
            idx, w, rows = self._tensors(src.device, src.dtype, ordered=src.layout == COLUMNS)
            native.regrid_ell(
                src.data, out.data, idx, w, n_src=self.n_src, n_tgt=self.n_tgt, k=self.k, n_lev=src.n_lev,
                src_pitch=src.pitch, out_pitch=out.pitch, layout=src.layout, prog=prog, n_stage=n_stage, tgt_mask=tgt_mask,
                padded=self.padded, **({} if rows is None else {"tgt_rows": rows}),
            )
        else:
            indptr, indices, data, rows = self._tensors(src.device, src.dtype, ordered=src.layout == COLUMNS)
            native.regrid_csr(
                src.data, out.data, indptr, indices, data, n_src=self.n_src, n_tgt=self.n_tgt, nnz=len(self.indices),
                n_lev=src.n_lev, src_pitch=src.pitch, out_pitch=out.pitch, layout=src.layout, prog=prog, n_stage=n_stage,
                tgt_mask=tgt_mask, **({} if rows is None else {"tgt_rows": rows}),
            )
        return out

    def bind(self, src: Stack, out: Stack | None = None, *, prog: torch.Tensor | None = None, n_stage: int = 0,
             tgt_mask: torch.Tensor | None = None):
        """``(launch, out)``: ``launch()`` repeats ``apply(src, out=out)`` with every argument converted once
        (``native.BoundCall``) — for a caller that regrids the same buffers again and again (one surface field per time step:
        BASELINE configs[1] is 3 us of kernel under ~10 us of launch, so what Python adds per call is what there is to save).
        The contents of ``src`` may change between calls, its storage may not.  Fixed-k plans in natural target order; anything
        else gets a closure over ``apply``."""
        assert src.n_pts == self.n_src, (src.n_pts, self.n_src)
        if out is None:
            out = src.new_like(n_pts=self.n_tgt, zero=False)
        else:
            self._check_out(src, out)
        plain = self.kind == "ell" and self.n_tgt > 0 and not (src.layout != COLUMNS and self._long_rows())
        if plain:
            idx, w, rows = self._tensors(src.device, src.dtype, ordered=src.layout == COLUMNS)
            if rows is None:
                call = native.bind_regrid_ell(src.data, out.data, idx, w, n_src=self.n_src, n_tgt=self.n_tgt, k=self.k, n_lev=src.n_lev,
                                              src_pitch=src.pitch, out_pitch=out.pitch, layout=src.layout, prog=prog, n_stage=n_stage,
                                              tgt_mask=tgt_mask, padded=self.padded)
                return call, out
        return (lambda: self.apply(src, prog=prog, n_stage=n_stage, tgt_mask=tgt_mask, out=out) and None), out

    def apply_many(self, stacks: list[Stack]) -> list[Stack]:
        """The gather over several source stacks of identical shape (variables / time steps on one grid) — fixed-k
        plans on column stacks run them in ONE launch (``atx_regrid_ell_batch``, grid.y = stack); anything else falls back to one launch per stack."""
        if not stacks:
            return []
        first = stacks[0]
        same = all((s.n_pts, s.n_lev, s.dtype, s.layout, s.pitch) == (first.n_pts, first.n_lev, first.dtype, first.layout, first.pitch)
                   for s in stacks)
        if self.kind != "ell" or not same or self.n_tgt == 0 or len(stacks) == 1 or (first.layout != COLUMNS and self._long_rows()):
            return [self.apply(s) for s in stacks]
        assert first.n_pts == self.n_src, (first.n_pts, self.n_src)
        outs = [first.new_like(n_pts=self.n_tgt, zero=False) for _ in stacks]
        idx, w, rows = self._tensors(first.device, first.dtype, ordered=first.layout == COLUMNS)
        native.regrid_ell_batch(
            [s.data for s in stacks], [o.data for o in outs], idx, w, n_src=self.n_src, n_tgt=self.n_tgt, k=self.k,
            n_lev=first.n_lev, src_pitch=first.pitch, out_pitch=outs[0].pitch, layout=first.layout, padded=self.padded,
            **({} if rows is None else {"tgt_rows": rows}),
        )
        return outs


def equal_count_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous partition of ``range(n)`` into ``world`` slices of (almost) equal length — for per-point work,
    whose cost is uniform; gathers use ``GatherPlan.bounds`` (balanced by traffic) instead."""
    return (n * rank) // world, (n * (rank + 1)) // world


import os as _os

ORDER_MIN_TARGETS = int(_os.environ.get("ATX_ORDER_MIN_TARGETS", 200_000))  # smaller target grids: their whole neighbourhood fits the caches whatever the order
ORDER_BLOCK_POINTS = int(_os.environ.get("ATX_ORDER_BLOCK_POINTS", 360))  # targets across a column block (measured on O1280 -> 0.25 degree, k = 16 and 8: 360 best, 160-256 within 3 %)
ORDER_MIN_K = int(_os.environ.get("ATX_ORDER_MIN_K", 5))  # (the ATX_ORDER_* variables exist for soak runs of the test suites with the ordered route forced on) rows of 1-4 neighbours LOSE 2-7 % to the scattered output rows; from 5 on the L2 hits win (k = 8: +5-9 %, k = 16: +9-12 %)


def target_order_for(latitudes, longitudes, k: int | None) -> np.ndarray | None:
    """The library's policy: ``column_block_order`` for fixed-k plans with at least ``ORDER_MIN_K`` neighbours per target, natural
    order otherwise (profiles/r03_column_blocks_experiment.log: the measured net effect on O1280 -> 0.25 degree, 137 levels)."""
    if k is None or k < ORDER_MIN_K or latitudes is None or longitudes is None:
        return None
    return column_block_order(latitudes, longitudes)


def column_block_order(latitudes: np.ndarray, longitudes: np.ndarray, block_points: int | None = None) -> np.ndarray | None:
    """A visiting order for ``GatherPlan.order_targets``: the target points in bands of longitude about ``block_points`` points
    wide, each band in the points' own order (row by row for a lat-lon or Gaussian grid).  ``None`` for small or degenerate grids.

    Why (round 4 corrected round 3's reading, profiles/r04_tile_orders_experiment.log, r04_xcd_stripes_experiment.log): the kernel deals
    its workgroups to the 8 XCDs in contiguous ranges, so in row-major order an XCD stays on ONE latitude belt — the polar ones, where
    hundreds of targets share a few source columns, finish early, the equatorial ones fetch every column from HBM and carry the launch.
    In full-height bands every XCD walks from pole to equator.  The number of bands is therefore kept a multiple of 4 (with 4 bands an
    XCD takes half a band, with 8 a whole one; 2, 3, 6 or 10 bands leave the XCD ranges across band boundaries and lose 1-8 %,
    profiles/r04_band_count_experiment.log).  The output rows of an ordered launch are written band by band instead of end to end, which
    costs a few per cent by itself: see ``target_order_for`` for when the library uses it."""
    block_points = ORDER_BLOCK_POINTS if block_points is None else block_points
    lat = np.asarray(latitudes, dtype=np.float64).reshape(-1)
    lon = np.asarray(longitudes, dtype=np.float64).reshape(-1)
    n = lat.size
    if n < ORDER_MIN_TARGETS or lon.size != n:
        return None
    lon = np.mod(lon, 360.0)
    lon_span, lat_span = float(lon.max() - lon.min()), float(lat.max() - lat.min())
    if lon_span <= 0.0 or lat_span <= 0.0:
        return None
    spacing = np.sqrt(lon_span * lat_span / n)  # mean point spacing in degrees
    n_bands = int(round(lon_span / (block_points * spacing)))
    if n_bands < 2:
        return None
    n_bands = max(4, 4 * int(round(n_bands / 4)))  # whole or half bands per XCD
    band = np.minimum(((lon - lon.min()) * (n_bands / lon_span)).astype(np.int64), n_bands - 1)
    return np.argsort(band, kind="stable").astype(np.int32)
