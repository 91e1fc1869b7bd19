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
import threading

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
    """Row pitch (elements) of a columns stack: n_lev rounded up to a 16-byte multiple, so that every column starts on a
    16-byte boundary and the kernels move 16-byte vectors — except for stacks thinner than one vector (1-3 float32
    levels, 1 float64 level), which are stored tight: padding a single surface field to 16 bytes per point would move 4x
    the bytes (measured on O1280 -> 0.25 degree, k = 4: 23.4 us padded, 12.7 us tight — the launch floor;
    profiles/r02_thin_stacks.log), and a tight 1-level column stack IS the field, contiguous."""
    per16 = 16 // torch.empty((), dtype=dtype).element_size()
    if n_lev < per16:
        return n_lev
    return (n_lev + per16 - 1) // per16 * per16


# ---- host <-> HBM staging ------------------------------------------------------------------
_PINNED_MIN_BYTES = 8 << 20    # below this a plain copy is as fast
_STAGE_BYTES = 256 << 20       # size of one pinned staging chunk (two are in flight)
_COPY_THREADS = max(1, min(8, (os.cpu_count() or 2) // 2))
_copy_pool = None
_copy_streams: dict[tuple, torch.cuda.Stream] = {}  # per (device, host thread): a prefetch thread and the consumer do not share one


def copy_pool():
    """Threads for host-side staging copies (numpy releases the GIL for large copies)."""
    global _copy_pool
    if _copy_pool is None:
        from concurrent.futures import ThreadPoolExecutor

        _copy_pool = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix="atx-stage")
    return _copy_pool


def _upload_rows(rows: list[np.ndarray], dst: torch.Tensor) -> None:
    """``dst[l, :] = rows[l]`` for a device tensor ``dst`` ``[n_lev, n_pts]`` (values cast to its dtype)."""
    n_lev, n_pts = dst.shape
    row_bytes = n_pts * dst.element_size()
    if not dst.is_cuda or n_lev * row_bytes < _PINNED_MIN_BYTES:
        np_dtype = np.float32 if dst.dtype == torch.float32 else np.float64
        host = np.empty((n_lev, n_pts), dtype=np_dtype)
        for l, r in enumerate(rows):
            host[l] = r
        dst.copy_(torch.from_numpy(host))
        return
    pool = copy_pool()
    key = (dst.device.index, threading.get_ident())
    stream = _copy_streams.get(key)
    if stream is None:
        if len(_copy_streams) >= 16:  # threads come and go
            _copy_streams.clear()
        stream = _copy_streams[key] = torch.cuda.Stream(dst.device)
    per_chunk = max(1, min(n_lev, _STAGE_BYTES // row_bytes))
    stages = [torch.empty((per_chunk, n_pts), dtype=dst.dtype, pin_memory=True) for _ in range(2 if n_lev > per_chunk else 1)]
    in_flight: list[torch.cuda.Event | None] = [None] * len(stages)
    stream.wait_stream(torch.cuda.current_stream(dst.device))  # dst may still be in use by earlier work
    for c, l0 in enumerate(range(0, n_lev, per_chunk)):
        l1 = min(n_lev, l0 + per_chunk)
        slot = c % len(stages)
        if in_flight[slot] is not None:
            in_flight[slot].synchronize()  # the DMA that last read this chunk has finished
        view = stages[slot].numpy()
        # numpy releases the GIL for large copies: the threads run at memory bandwidth
        list(pool.map(lambda l: np.copyto(view[l - l0], rows[l], casting="unsafe"), range(l0, l1)))
        with torch.cuda.stream(stream):
            dst[l0:l1].copy_(stages[slot][: l1 - l0], non_blocking=True)
            in_flight[slot] = torch.cuda.Event()
            in_flight[slot].record()
    torch.cuda.current_stream(dst.device).wait_stream(stream)
    for ev in in_flight:  # the staging chunks go back to the host allocator only once the DMA engine is done with them
        if ev is not None:
            ev.synchronize()


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

    @property
    def layout_name(self) -> str:
        return "columns" if self.layout == COLUMNS else "fields"

    def __repr__(self) -> str:
        return f"Stack({self.n_lev} levels x {self.n_pts} points, {self.layout_name}, {self.dtype}, pitch={self.pitch})"

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

        The host side is field-major (the reference's unit: one array per field).  Fields are copied by
        a few threads into pinned staging chunks, each chunk goes to HBM by an asynchronous DMA on a
        copy stream while the next one is being filled, and the stack is re-laid out on the device by
        ``atx_relayout``.
        """
        dev = device() if dev is None else dev
        rows = [np.asarray(a).reshape(-1) for a in arrays]
        if dtype is None:
            dtype = torch.float32 if all(r.dtype == np.float32 for r in rows) else torch.float64
        n_pts = rows[0].size
        assert all(r.size == n_pts for r in rows), "fields of one stack must share a grid"
        staged = cls(torch.empty((len(rows), n_pts), dtype=dtype, device=dev), n_pts, len(rows), FIELDS)
        _upload_rows(rows, staged.data)
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
        if fm.data.is_cuda and fm.data.numel() * fm.data.element_size() >= _PINNED_MIN_BYTES:
            host = torch.empty((self.n_lev, self.n_pts), dtype=self.dtype, pin_memory=True)  # DMA at PCIe rate, one copy
            host.copy_(fm.data[:, : self.n_pts], non_blocking=True)
            torch.cuda.current_stream(fm.device).synchronize()
            return host.numpy()
        return fm.data[:, : self.n_pts].cpu().numpy()  # a FIELDS stack has pitch n_pts: already contiguous
