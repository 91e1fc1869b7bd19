"""Stacks: batches of same-grid fields resident in HBM.

The reference moves data one 2-D field at a time (``field.to_numpy()``,
R: fields.py:178-202) through a Python loop (R: filter.py:188-196,
filters/fields/regrid.py:204-208).  Here consecutive same-grid fields of a
FieldList live together in one device tensor, a *stack*, and every filter is one
kernel launch over the whole stack.

Native layout: **columns** — ``data[p, l]`` with all levels of grid point ``p``
contiguous (pitch rounded up to 16 bytes).  A neighbour read of the regrid
gather is then one contiguous run of ``n_lev`` values, which makes the HBM
traffic of the O1280 -> 0.25 degree regrid 1.05-1.09x the algorithmic bytes
instead of 1.5-2.5x for field-major storage (measured).  The
field-major view the reference API needs (one field = one array) is produced on
demand by the ``atx_relayout`` kernel.
"""

from __future__ import annotations

import os

import numpy as np
import torch

from . import native

COLUMNS = native.COLUMNS
FIELDS = native.FIELDS


def device() -> torch.device:
    """The HBM device of this process: ``cuda:LOCAL_RANK`` (one process per GPU)."""
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible: this package has no CPU compute path")
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())


def column_pitch(n_lev: int, dtype: torch.dtype) -> int:
    """Row pitch (elements) of a columns stack: n_lev rounded up to a 16-byte multiple."""
    per16 = 16 // torch.empty((), dtype=dtype).element_size()
    return (n_lev + per16 - 1) // per16 * per16


class Stack:
    """``n_lev`` fields on one grid of ``n_pts`` points, in HBM."""

    __slots__ = ("data", "n_pts", "n_lev", "layout")

    def __init__(self, data: torch.Tensor, n_pts: int, n_lev: int, layout: int = COLUMNS) -> None:
        assert data.dim() == 2 and data.stride(1) == 1, "a stack is a pitched 2-D tensor"
        if layout == COLUMNS:
            assert data.shape[0] == n_pts and data.shape[1] >= n_lev, (tuple(data.shape), n_pts, n_lev)
        else:
            assert data.shape[0] == n_lev and data.shape[1] >= n_pts, (tuple(data.shape), n_pts, n_lev)
        self.data = data
        self.n_pts = n_pts
        self.n_lev = n_lev
        self.layout = layout

    # ---- geometry --------------------------------------------------------------------
    @property
    def pitch(self) -> int:
        return self.data.stride(0)

    @property
    def dtype(self) -> torch.dtype:
        return self.data.dtype

    @property
    def device(self) -> torch.device:
        return self.data.device

    def __repr__(self) -> str:
        name = "columns" if self.layout == COLUMNS else "fields"
        return f"Stack({self.n_lev} levels x {self.n_pts} points, {name}, {self.dtype}, pitch={self.pitch})"

    # ---- construction ----------------------------------------------------------------
    @classmethod
    def empty(cls, n_pts: int, n_lev: int, dtype: torch.dtype, dev=None, layout: int = COLUMNS, zero: bool = False) -> "Stack":
        dev = device() if dev is None else dev
        make = torch.zeros if zero else torch.empty
        if layout == COLUMNS:
            data = make((n_pts, column_pitch(n_lev, dtype)), dtype=dtype, device=dev)
        else:
            data = make((n_lev, n_pts), dtype=dtype, device=dev)
        return cls(data, n_pts, n_lev, layout)

    @classmethod
    def from_fields(cls, arrays, dtype: torch.dtype | None = None, dev=None, layout: int = COLUMNS) -> "Stack":
        """Upload ``n_lev`` flattened host fields (each ``[n_pts]``) as one stack.

        The host array is field-major (the reference's unit); it is copied to HBM
        once and re-laid out on the device by ``atx_relayout``.
        """
        dev = device() if dev is None else dev
        host = np.ascontiguousarray(np.stack([np.asarray(a).reshape(-1) for a in arrays], axis=0))
        t = torch.from_numpy(host)
        if dtype is not None:
            t = t.to(dtype)
        staged = cls(t.to(dev), host.shape[1], host.shape[0], FIELDS)
        return staged if layout == FIELDS else staged.to_layout(COLUMNS)

    # ---- layout ----------------------------------------------------------------------
    def to_layout(self, layout: int) -> "Stack":
        if layout == self.layout:
            return self
        out = Stack.empty(self.n_pts, self.n_lev, self.dtype, self.device, layout, zero=(layout == COLUMNS))
        native.relayout(
            self.data, out.data, n_pts=self.n_pts, n_lev=self.n_lev, src_pitch=self.pitch, dst_pitch=out.pitch,
            src_layout=self.layout, dst_layout=layout,
        )
        return out

    def new_like(self, n_pts: int | None = None, n_lev: int | None = None, zero: bool = False) -> "Stack":
        return Stack.empty(
            self.n_pts if n_pts is None else n_pts, self.n_lev if n_lev is None else n_lev, self.dtype, self.device,
            self.layout, zero=zero,
        )

    # ---- host views ------------------------------------------------------------------
    def level_view(self, level: int) -> torch.Tensor:
        """Strided device view of one field (no copy)."""
        if self.layout == COLUMNS:
            return self.data[:, level]
        return self.data[level, : self.n_pts]

    def level_numpy(self, level: int) -> np.ndarray:
        """One field as a flat host array — the reference's ``to_numpy(flatten=True)`` (implies D2H)."""
        if self.layout == FIELDS:
            return self.data[level, : self.n_pts].cpu().numpy()
        row = torch.empty((1, self.n_pts), dtype=self.dtype, device=self.device)
        native.relayout(self.data[:, level : level + 1], row, n_pts=self.n_pts, n_lev=1, src_pitch=self.pitch,
                        dst_pitch=self.n_pts, src_layout=COLUMNS, dst_layout=FIELDS)
        return row[0].cpu().numpy()

    def numpy(self) -> np.ndarray:
        """All fields, field-major ``[n_lev, n_pts]``, on the host."""
        fm = self.to_layout(FIELDS)
        return fm.data[:, : self.n_pts].cpu().numpy()  # a FIELDS stack has pitch n_pts: already contiguous
