"""ctypes binding of libatx — the only compute path of this package.

There is no CPU fallback: if ``lib/libatx.so`` is missing, or a tensor handed to
one of the wrappers does not live in HBM, the call raises.  Return codes of the
C ABI (``include/atx.h``) are mapped to the exception types the reference raises
on this path: ``ValueError`` for bad configuration, ``AssertionError`` for shape
mismatches (R: filters/fields/regrid.py:377-378), ``NotImplementedError`` for
unsupported combinations (R: regrid.py:332-335).
"""

from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_size_t, c_void_p
from typing import Any

import numpy as np
import torch

LIB_NAME = "libatx.so"
LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")

# atx_dtype / atx_layout / atx_cmp / atx_op / atx_red (include/atx.h)
F32, F64 = 0, 1
COLUMNS, FIELDS = 0, 1
CMP_GT, CMP_LT, CMP_EQ, CMP_NE, CMP_GE, CMP_LE, CMP_NOTNAN, CMP_ISNAN = range(8)
(
    OP_COPY,
    OP_AFFINE,
    OP_AFFINE_INV,
    OP_MUL,
    OP_DIV,
    OP_CLIP,
    OP_IMPUTE_NAN,
    OP_EXP,
    OP_LOG,
    OP_SET_NAN,
) = range(10)
RED_MIN, RED_MAX, RED_NANCOUNT, RED_MINMAX = range(4)
(COMB_SNOW_DEPTH_M, COMB_SNOW_COVER, COMB_COS_SIN, COMB_ATAN2, COMB_W_TO_WZ, COMB_WZ_TO_W, COMB_SUM, COMB_SUB, COMB_XY_TO_POLAR,
 COMB_POLAR_TO_XY, COMB_OPERA_CLIP, COMB_OPERA_PREPROCESS, COMB_ORAS6, COMB_LOOKUP, COMB_R_TO_D, COMB_D_TO_R, COMB_Q_TO_R,
 COMB_R_TO_Q) = range(18)
# what a level of a COMB_ORAS6 stack holds (ATX_ORAS6_* of atx.h)
ORAS6_KEEP, ORAS6_ZERO, ORAS6_TEMPERATURE, ORAS6_CELSIUS, ORAS6_HEAT, ORAS6_SURFACE = range(6)
COMB_DEGREES = 1
COMB_MAX_INPUTS = 8

OK, EINVAL, ESHAPE, ENOTIMPL, EHIP, EALIGN, EWORKSPACE, ECOMM = 0, -1, -2, -3, -4, -5, -6, -7
COMM_ID_BYTES = 128

# numpy layout of `atx_level_op` (24 bytes)
LEVEL_OP_DTYPE = np.dtype([("op", "<i4"), ("use_mask", "<i4"), ("p0", "<f8"), ("p1", "<f8")])

# every symbol include/atx.h declares: name -> (restype, argtypes)
SIGNATURES: dict[str, tuple[Any, list[Any]]] = {
    "atx_version": (c_int, []),
    "atx_last_error": (c_char_p, []),
    "atx_strerror": (c_char_p, [c_int]),
    "atx_device_count": (c_int, []),
    "atx_set_tuning": (c_int, [c_int]),
    "atx_regrid_ell": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int64, c_int64, c_int64, c_int, c_int, c_int32,
         c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p],
    ),
    "atx_regrid_ell_batch": (
        c_int,
        [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int64, c_int64, c_int64, c_int, c_int,
         c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p],
    ),
    "atx_regrid_ell_ordered": (
        c_int,
        [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int64, c_int64, c_int64, c_int, c_int, c_int32,
         c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p],
    ),
    "atx_regrid_csr": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int,
         c_int, c_void_p, c_int32, c_void_p, c_void_p],
    ),
    "atx_regrid_csr_ordered": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int,
         c_int, c_void_p, c_int32, c_void_p, c_void_p],
    ),
    "atx_check_indices": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "atx_pointwise_stack": (
        c_int,
        [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p],
    ),
    "atx_combine_stack": (
        c_int,
        [c_int, POINTER(c_void_p), c_int32, POINTER(c_void_p), c_int32, c_int64, c_int64, c_int64, c_int, c_int, c_void_p,
         c_int32, c_void_p],
    ),
    "atx_mask_build": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_double, c_int, c_void_p]),
    "atx_mask_count": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "atx_mask_to_index_workspace": (c_size_t, [c_int64]),
    "atx_mask_to_index": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "atx_knn_workspace_bytes": (c_size_t, [c_int64]),
    "atx_knn_build": (c_int, [c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "atx_knn_query": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "atx_cutout_inside": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p]),
    "atx_stream_copy": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "atx_vector_program": (c_int64, [c_void_p, c_int32, c_int64, c_int, c_void_p, c_int64]),
    "atx_reduce_workspace": (c_size_t, []),
    "atx_reduce": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "atx_relayout": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "atx_reduce_stack": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "atx_select_levels": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "atx_comm_version": (c_int, []),
    "atx_comm_unique_id": (c_int, [c_void_p]),
    "atx_comm_init": (c_int, [POINTER(c_void_p), c_int32, c_int32, c_void_p]),
    "atx_comm_destroy": (c_int, [c_void_p]),
    "atx_comm_rank": (c_int, [c_void_p]),
    "atx_comm_world": (c_int, [c_void_p]),
    "atx_bcast": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "atx_all_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "atx_exchange": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p), POINTER(c_int64), c_void_p]),
    "atx_gather_shards": (c_int, [c_void_p, c_void_p, POINTER(c_int64), c_void_p]),
}

_lib: ctypes.CDLL | None = None


class AtxError(RuntimeError):
    """HIP runtime failure inside libatx."""


def lib_path() -> str:
    """``lib/libatx.so`` next to this file; ``ATX_LIBRARY`` overrides it (A/B builds of the kernels)."""
    return os.environ.get("ATX_LIBRARY") or os.path.join(LIB_DIR, LIB_NAME)


def load() -> ctypes.CDLL:
    """Load libatx.so (once) and declare every prototype.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run `python -c \"import __graft_entry__ as g; "
            "g.build()\"` (hipcc --offload-arch=gfx950) — this package has no CPU fallback."
        )
    handle = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return handle


def _raise(code: int, fn: str) -> None:
    handle = load()
    msg = handle.atx_last_error().decode() or handle.atx_strerror(code).decode()
    if code == EINVAL:
        raise ValueError(msg)
    if code == ESHAPE:
        raise AssertionError(msg)
    if code == ENOTIMPL:
        raise NotImplementedError(msg)
    if code == ECOMM:
        raise AtxError(f"{fn}: {msg}")
    raise AtxError(f"{fn}: {msg} (code {code})")


def _call(fn: str, *args: Any) -> None:
    code = getattr(load(), fn)(*args)
    if code != OK:
        _raise(code, fn)


def _ptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"libatx works on HBM-resident tensors only, got a {t.device} tensor (no CPU fallback)")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.float64:
        return F64
    raise ValueError(f"libatx supports float32 / float64 stacks, got {dtype}")


def version() -> int:
    return load().atx_version()


def device_count() -> int:
    return load().atx_device_count()


def set_tuning(tile: int) -> None:
    _call("atx_set_tuning", int(tile))


# --------------------------------------------------------------------------------
# tensor-level wrappers (these are what the filters call)
# --------------------------------------------------------------------------------
ELL_PADDED = 1


def _program_companions(prog, dtype) -> tuple[int | None, int | None]:
    """``(vec_prog device pointer, host_prog host pointer)`` that ``level_program`` attached to a program tensor, if any."""
    if prog is None:
        return None, None
    vec = getattr(prog, "vec_prog", {}).get(dtype)
    host = getattr(prog, "host_prog", None)
    return _ptr(vec), (None if host is None else host.ctypes.data)


def regrid_ell(src, out, idx, w, *, n_src, n_tgt, k, n_lev, src_pitch, out_pitch, layout, prog=None, n_stage=0,
               tgt_mask=None, padded: bool = False, tgt_rows=None) -> None:
    """out[t, l] = sum_j w[t, j] * src[idx[t, j], l]; ``w is None`` -> pure k = 1 gather;
    ``padded``: negative indices are absent entries of padded ragged rows (ATX_ELL_PADDED);
    ``tgt_rows`` (device int32 permutation): ordered traversal — table row t is output row ``tgt_rows[t]``
    (``atx_regrid_ell_ordered``; column stacks only)."""
    if tgt_rows is not None:
        return regrid_ell_batch([src], [out], idx, w, n_src=n_src, n_tgt=n_tgt, k=k, n_lev=n_lev, src_pitch=src_pitch, out_pitch=out_pitch,
                                layout=layout, prog=prog, n_stage=n_stage, tgt_mask=tgt_mask, padded=padded, tgt_rows=tgt_rows)
    _call("atx_regrid_ell", *_regrid_ell_args(src, out, idx, w, n_src, n_tgt, k, n_lev, src_pitch, out_pitch, layout, prog, n_stage, tgt_mask,
                                              padded), _stream())


def _regrid_ell_args(src, out, idx, w, n_src, n_tgt, k, n_lev, src_pitch, out_pitch, layout, prog, n_stage, tgt_mask, padded) -> tuple:
    """The arguments of ``atx_regrid_ell`` up to (not including) the stream, checked and converted."""
    assert src.dtype == out.dtype, (src.dtype, out.dtype)
    assert idx.dtype == torch.int32
    if w is not None:
        assert w.dtype == src.dtype, (w.dtype, src.dtype)
    vec, host = _program_companions(prog, src.dtype)
    return (_ptr(src), _ptr(out), _ptr(idx), _ptr(w), n_src, n_tgt, k, n_lev, src_pitch, out_pitch, dtype_code(src.dtype), layout,
            ELL_PADDED if padded else 0, _ptr(prog), vec, host, n_stage, _ptr(tgt_mask))


class BoundCall:
    """One libatx entry point with its arguments checked and converted ONCE, for call sites that repeat the same launch on the same
    buffers (a one-field regrid per time step is launch-bound: its kernel runs 3 us, the Python marshalling of 19 arguments costs
    more than that).  ``__call__`` enqueues on the stream that was current when the call was bound, or on ``stream=`` given then.
    The object keeps the tensors it points into alive."""

    __slots__ = ("_fn", "_name", "_args", "_keep")

    def __init__(self, name: str, args: tuple, keep: tuple, stream: int | None = None) -> None:
        self._fn = getattr(load(), name)
        self._name = name
        self._args = (*args, _stream() if stream is None else stream)
        self._keep = keep

    def __call__(self) -> None:
        code = self._fn(*self._args)
        if code != OK:
            _raise(code, self._name)


def bind_regrid_ell(src, out, idx, w, *, n_src, n_tgt, k, n_lev, src_pitch, out_pitch, layout, prog=None, n_stage=0, tgt_mask=None,
                    padded: bool = False, stream: int | None = None) -> BoundCall:
    """``regrid_ell`` (natural target order) as a ``BoundCall``."""
    args = _regrid_ell_args(src, out, idx, w, n_src, n_tgt, k, n_lev, src_pitch, out_pitch, layout, prog, n_stage, tgt_mask, padded)
    return BoundCall("atx_regrid_ell", args, (src, out, idx, w, prog, tgt_mask), stream)


def regrid_ell_batch(srcs, outs, idx, w, *, n_src, n_tgt, k, n_lev, src_pitch, out_pitch, layout, prog=None, n_stage=0,
                     tgt_mask=None, padded: bool = False, tgt_rows=None) -> None:
    """``regrid_ell`` over several stacks of identical shape in one launch (``atx_regrid_ell_batch``; with ``tgt_rows``:
    ``atx_regrid_ell_ordered``): ``srcs`` / ``outs`` are sequences of device tensors."""
    assert len(srcs) == len(outs) >= 1
    assert idx.dtype == torch.int32
    dtype = srcs[0].dtype
    assert all(t.dtype == dtype for t in srcs) and all(t.dtype == dtype for t in outs) and (w is None or w.dtype == dtype)
    n = len(srcs)
    src_ptrs = (c_void_p * n)(*[_ptr(t) for t in srcs])
    out_ptrs = (c_void_p * n)(*[_ptr(t) for t in outs])
    vec, host = _program_companions(prog, dtype)
    if tgt_rows is not None:
        assert tgt_rows.dtype == torch.int32 and tgt_rows.numel() >= n_tgt
        _call(
            "atx_regrid_ell_ordered", ctypes.cast(src_ptrs, c_void_p), ctypes.cast(out_ptrs, c_void_p), n, _ptr(idx), _ptr(w), _ptr(tgt_rows),
            n_src, n_tgt, k, n_lev, src_pitch, out_pitch, dtype_code(dtype), layout, ELL_PADDED if padded else 0, _ptr(prog), vec, host,
            n_stage, _ptr(tgt_mask), _stream(),
        )
        return
    _call(
        "atx_regrid_ell_batch", ctypes.cast(src_ptrs, c_void_p), ctypes.cast(out_ptrs, c_void_p), n, _ptr(idx), _ptr(w), n_src,
        n_tgt, k, n_lev, src_pitch, out_pitch, dtype_code(dtype), layout, ELL_PADDED if padded else 0, _ptr(prog), vec, host, n_stage,
        _ptr(tgt_mask), _stream(),
    )


def regrid_csr(src, out, indptr, indices, data, *, n_src, n_tgt, nnz, n_lev, src_pitch, out_pitch, layout, prog=None,
               n_stage=0, tgt_mask=None, tgt_rows=None) -> None:
    """out[t, l] = sum over the CSR row t of data * src[indices, l] (scipy csr_matvec order); with ``tgt_rows`` (device int32
    permutation) CSR row t is output row ``tgt_rows[t]`` (``atx_regrid_csr_ordered``; column stacks only)."""
    assert src.dtype == out.dtype == data.dtype, (src.dtype, out.dtype, data.dtype)
    assert indptr.dtype == torch.int32 and indices.dtype == torch.int32
    if tgt_rows is not None:
        assert tgt_rows.dtype == torch.int32 and tgt_rows.numel() >= n_tgt
        _call(
            "atx_regrid_csr_ordered", _ptr(src), _ptr(out), _ptr(indptr), _ptr(indices), _ptr(data), _ptr(tgt_rows), n_src, n_tgt, nnz, n_lev,
            src_pitch, out_pitch, dtype_code(src.dtype), layout, _ptr(prog), n_stage, _ptr(tgt_mask), _stream(),
        )
        return
    _call(
        "atx_regrid_csr", _ptr(src), _ptr(out), _ptr(indptr), _ptr(indices), _ptr(data), n_src, n_tgt, nnz, n_lev,
        src_pitch, out_pitch, dtype_code(src.dtype), layout, _ptr(prog), n_stage, _ptr(tgt_mask), _stream(),
    )


def check_indices(idx, n_src: int) -> int:
    """Number of entries of ``idx`` outside [0, n_src) (synchronises)."""
    n_bad = torch.zeros(1, dtype=torch.int64, device=idx.device)
    _call("atx_check_indices", _ptr(idx), idx.numel(), n_src, _ptr(n_bad), _stream())
    return int(n_bad.item())


_PROGRAM_CACHE: dict[Any, torch.Tensor] = {}
_PROGRAM_CACHE_SIZE = 64


def level_program(stages: list[list[tuple[int, int, float, float]]], device, cache: bool = False) -> torch.Tensor:
    """Device copy of a per-level program: ``stages[s][l] = (op, use_mask, p0, p1)``.

    ``cache=True`` (the filters' call): programs are remembered by value — a pipeline applies the same program to every
    FieldList of a job, and rebuilding it (three small host-to-device copies and the per-vector tables) costs about as
    much host time as the launch it drives.  Cached programs are shared: treat them as read-only."""
    key = None
    if cache:
        key = (tuple(tuple(tuple(entry) for entry in stage) for stage in stages), str(device))
        hit = _PROGRAM_CACHE.get(key)
        if hit is not None:
            return hit
    raw = _build_level_program(stages, device)
    if key is not None:
        if len(_PROGRAM_CACHE) >= _PROGRAM_CACHE_SIZE:
            _PROGRAM_CACHE.pop(next(iter(_PROGRAM_CACHE)))
        _PROGRAM_CACHE[key] = raw
    return raw


def _build_level_program(stages, device) -> torch.Tensor:
    n_stage = len(stages)
    n_lev = len(stages[0])
    host = np.zeros(n_stage * n_lev, dtype=LEVEL_OP_DTYPE)
    for s, stage in enumerate(stages):
        assert len(stage) == n_lev
        for l, (op, use_mask, p0, p1) in enumerate(stage):
            host[s * n_lev + l] = (op, use_mask, p0, p1)
    raw = torch.from_numpy(host.view(np.uint8).copy()).to(device)
    # the host copy and the per-vector forms (atx_vector_program, host side) ride along: kernels given them need no
    # per-workgroup set-up (atx_regrid_ell: host_prog / vec_prog; atx_pointwise_stack: vec_prog)
    raw.host_prog = host
    raw.vec_prog = {}
    if raw.is_cuda:
        lib = load()
        for tdtype, code in ((torch.float32, F32), (torch.float64, F64)):
            n_entries = lib.atx_vector_program(host.ctypes.data, n_stage, n_lev, code, None, 0)
            table = np.zeros(n_entries, dtype=LEVEL_OP_DTYPE)
            got = lib.atx_vector_program(host.ctypes.data, n_stage, n_lev, code, table.ctypes.data, n_entries)
            if got != n_entries:
                raise AtxError(f"atx_vector_program: {lib.atx_last_error().decode()}")
            raw.vec_prog[tdtype] = torch.from_numpy(table.view(np.uint8).copy()).to(device)
    return raw


def pointwise_stack(x, y, *, n_pts, n_lev, x_pitch, y_pitch, layout, prog, n_stage, point_mask=None) -> None:
    assert x.dtype == y.dtype
    vec, host = _program_companions(prog, x.dtype)
    _call(
        "atx_pointwise_stack", _ptr(x), _ptr(y), n_pts, n_lev, x_pitch, y_pitch, dtype_code(x.dtype), layout,
        _ptr(prog), vec, host, n_stage, _ptr(point_mask), _stream(),
    )


def combine_stack(op: int, inputs, outputs, *, n_pts, n_lev, pitch, layout, level_param=None, flags: int = 0) -> None:
    """Multi-input per-point operator over same-shape stacks (``atx_combine_stack``).  ``level_param``: a float64 value per level, except
    for ``COMB_LOOKUP`` where it is the table (its length, then its values); the second input of ``COMB_ORAS6`` is one field, not a stack."""
    dtype = inputs[0].dtype
    assert all(t.dtype == dtype for t in list(inputs) + list(outputs))
    ins = (c_void_p * len(inputs))(*[_ptr(t) for t in inputs])
    outs = (c_void_p * len(outputs))(*[_ptr(t) for t in outputs])
    if op == COMB_ORAS6:
        assert inputs[1].is_contiguous() and inputs[1].numel() >= n_pts
    if level_param is not None:
        assert level_param.dtype == torch.float64 and level_param.is_contiguous()
        assert level_param.numel() >= (2 if op == COMB_LOOKUP else n_lev)
    _call("atx_combine_stack", op, ins, len(inputs), outs, len(outputs), n_pts, n_lev, pitch, dtype_code(dtype), layout,
          _ptr(level_param), flags, _stream())


def mask_build(m, mask, *, n, stride=1, cmp, threshold=0.0) -> None:
    assert mask.dtype == torch.uint8
    _call("atx_mask_build", _ptr(m), stride, _ptr(mask), n, cmp, float(threshold), dtype_code(m.dtype), _stream())


def mask_count(mask, n: int | None = None) -> int:
    n = mask.numel() if n is None else n
    count = torch.zeros(1, dtype=torch.int64, device=mask.device)
    _call("atx_mask_count", _ptr(mask), n, _ptr(count), _stream())
    return int(count.item())


def mask_to_index(mask, n: int | None = None) -> torch.Tensor:
    """Ascending int32 positions of the set mask bytes (stable compaction; synchronises)."""
    n = mask.numel() if n is None else n
    slot = _Scratch.of(mask.device)
    workspace = slot.room(max(load().atx_mask_to_index_workspace(n), 16))
    index = torch.empty(max(n, 1), dtype=torch.int32, device=mask.device)  # the caller keeps (a view of) this one
    # the count lands in the pinned host cell straight from the scan kernel
    _call("atx_mask_to_index", _ptr(mask), n, _ptr(index), slot.host_count.data_ptr(), _ptr(workspace), workspace.numel(), _stream())
    slot.wait()
    return index[: int(slot.host_count[0])]


class KnnIndex:
    """A device k-NN index over source points (``atx_knn_build`` / ``atx_knn_query``)."""

    def __init__(self, src_xyz: torch.Tensor) -> None:
        assert src_xyz.dtype == torch.float64 and src_xyz.dim() == 2 and src_xyz.shape[1] == 3 and src_xyz.is_contiguous()
        self.n_src = src_xyz.shape[0]
        nbytes = load().atx_knn_workspace_bytes(self.n_src)
        if nbytes == 0:
            raise ValueError(f"cannot index {self.n_src} source points")
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=src_xyz.device)
        shift = (-self.workspace.data_ptr()) % 256
        self._ws = self.workspace[shift: shift + nbytes]
        _call("atx_knn_build", _ptr(src_xyz), self.n_src, _ptr(self._ws), nbytes, _stream())

    def query(self, tgt_xyz: torch.Tensor, k: int) -> tuple[torch.Tensor, torch.Tensor]:
        """``(indices int32 [n, k], squared distances float64 [n, k])``, nearest first."""
        assert tgt_xyz.dtype == torch.float64 and tgt_xyz.dim() == 2 and tgt_xyz.shape[1] == 3 and tgt_xyz.is_contiguous()
        n = tgt_xyz.shape[0]
        idx = torch.empty((n, k), dtype=torch.int32, device=tgt_xyz.device)
        d2 = torch.empty((n, k), dtype=torch.float64, device=tgt_xyz.device)
        _call("atx_knn_query", _ptr(self._ws), self.n_src, _ptr(tgt_xyz), n, k, _ptr(idx), _ptr(d2), _stream())
        return idx, d2


def cutout_inside(global_xyz: torch.Tensor, lam_xyz: torch.Tensor, neighbours: torch.Tensor) -> torch.Tensor:
    """uint8 [n]: the ray through each global point hits a triangle of its nearest LAM points (``atx_cutout_inside``)."""
    assert global_xyz.dtype == lam_xyz.dtype == torch.float64 and neighbours.dtype == torch.int32
    n, k = neighbours.shape
    inside = torch.empty(max(n, 1), dtype=torch.uint8, device=global_xyz.device)
    _call("atx_cutout_inside", _ptr(global_xyz), n, _ptr(lam_xyz), lam_xyz.shape[0], _ptr(neighbours), k, _ptr(inside), _stream())
    return inside[:n]


class _Scratch:
    """Per-(device, stream) slots for the small synchronous calls (``reduce``, ``reduce_stack``, ``mask_to_index``): pinned HOST
    cells the kernels write their result into directly (pinned memory is device-visible at the same address), the zeroed
    workspace of the ticketed reduction, the event the host waits on and a growing scratch buffer.  Such a call is then its
    launch(es) and ONE event wait — no allocation, no fill, no initialisation launch, no copy back, no blocking ``.item()``
    (``reduce`` of one 26 MB field: 45 us -> see tools/small_case_bench.py).  Slots are per (device, stream, host thread)."""

    _slots: dict = {}

    def __init__(self, device) -> None:
        self.result = torch.zeros(2, dtype=torch.float64, device=device)
        self.host = torch.zeros(2, dtype=torch.float64).pin_memory()
        self.host_count = torch.zeros(1, dtype=torch.int64).pin_memory()
        self.reduce_ws = torch.empty(load().atx_reduce_workspace() // 8 + 1, dtype=torch.int64, device=device)
        self.event = torch.cuda.Event()
        self.workspace = torch.empty(4096, dtype=torch.uint8, device=device)

    @classmethod
    def of(cls, device) -> "_Scratch":
        # per host thread as well: a producer thread (prefetch) and the consumer may enqueue on the same stream
        key = (device.index if device.index is not None else torch.cuda.current_device(), _stream(), threading.get_ident())
        slot = cls._slots.get(key)
        if slot is None:
            if len(cls._slots) >= 64:  # streams come and go
                cls._slots.clear()
            slot = cls._slots[key] = cls(device)
        return slot

    def wait(self) -> None:
        self.event.record()
        self.event.synchronize()

    def room(self, n_bytes: int) -> torch.Tensor:
        if self.workspace.numel() < n_bytes:
            self.workspace = torch.empty(max(n_bytes, 2 * self.workspace.numel()), dtype=torch.uint8, device=self.workspace.device)
        return self.workspace


# ATX_REDUCE_WORKSPACE=0 takes the library's route without a workspace (per-workgroup atomics on a device cell + a copy back).
_REDUCE_TICKET = os.environ.get("ATX_REDUCE_WORKSPACE", "1") == "1"


def _reduce_ws(slot: "_Scratch"):
    if _REDUCE_TICKET:
        return _ptr(slot.reduce_ws), slot.reduce_ws.numel() * 8, slot.host.data_ptr()
    return None, 0, _ptr(slot.result)


def _reduction_result(slot: _Scratch, red: int):
    if not _REDUCE_TICKET:
        slot.host.copy_(slot.result, non_blocking=True)
    slot.wait()
    if red == RED_MINMAX:
        return float(slot.host[0]), float(slot.host[1])  # both from one pass, one wait
    return float(slot.host[0])


def reduce(x, red: int, n: int | None = None):
    """``atx_reduce``; ``RED_MINMAX`` returns ``(minimum, maximum)`` from one pass."""
    n = x.numel() if n is None else n
    slot = _Scratch.of(x.device)
    ws, ws_bytes, result = _reduce_ws(slot)
    _call("atx_reduce", _ptr(x), n, red, result, dtype_code(x.dtype), ws, ws_bytes, _stream())
    return _reduction_result(slot, red)


def reduce_stack(x, red: int, *, n_pts: int, n_lev: int, pitch: int, layout: int):
    """``atx_reduce`` over the elements of a pitched stack (padding excluded); ``RED_MINMAX``: ``(minimum, maximum)``."""
    slot = _Scratch.of(x.device)
    ws, ws_bytes, result = _reduce_ws(slot)
    _call("atx_reduce_stack", _ptr(x), n_pts, n_lev, pitch, red, result, dtype_code(x.dtype), layout, ws, ws_bytes, _stream())
    return _reduction_result(slot, red)


def select_levels(src, dst, level_map, *, n_pts, n_src_lev, src_pitch, dst_pitch, layout) -> None:
    """dst level j = src level ``level_map[j]`` (negative: leave dst level j alone); ``level_map`` is a host sequence."""
    assert src.dtype == dst.dtype
    lm = (ctypes.c_int32 * len(level_map))(*[int(l) for l in level_map])
    _call("atx_select_levels", _ptr(src), _ptr(dst), ctypes.cast(lm, c_void_p), len(level_map), n_pts, n_src_lev,
          src_pitch, dst_pitch, dtype_code(src.dtype), layout, _stream())


def stream_copy(src, dst) -> None:
    """``atx_stream_copy``: the library's fixed reference streaming copy (calibration / ceiling measurements)."""
    n_bytes = src.numel() * src.element_size()
    assert dst.numel() * dst.element_size() == n_bytes and src.is_contiguous() and dst.is_contiguous()
    _call("atx_stream_copy", _ptr(src), _ptr(dst), n_bytes, _stream())


class Comm:
    """An RCCL communicator through the C ABI (``atx_comm_*``): one per process, bound to the current device.

    ``Comm.unique_id()`` on rank 0, the 128 bytes handed to every rank out of band, then ``Comm(world, rank, id)``
    everywhere (collective).  Calls enqueue on torch's current stream."""

    def __init__(self, world: int, rank: int, unique_id: bytes) -> None:
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError(f"an RCCL unique id has {COMM_ID_BYTES} bytes, got {len(unique_id)}")
        handle = c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        _call("atx_comm_init", ctypes.byref(handle), int(world), int(rank), ctypes.cast(buf, c_void_p))
        self._handle = handle
        self.world, self.rank = int(world), int(rank)

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        _call("atx_comm_unique_id", ctypes.cast(buf, c_void_p))
        return buf.raw

    @staticmethod
    def rccl_version() -> int:
        v = load().atx_comm_version()
        if v < 0:
            _raise(v, "atx_comm_version")
        return v

    def bcast(self, t: torch.Tensor, root: int) -> None:
        """``t`` (contiguous, in HBM) of rank ``root`` onto every rank, in place."""
        assert t.is_contiguous()
        _call("atx_bcast", self._handle, _ptr(t), t.numel() * t.element_size(), int(root), _stream())

    def all_gather(self, mine: torch.Tensor, everyone: torch.Tensor) -> None:
        """``mine`` (contiguous) of every rank into ``everyone`` (``world`` times its size): slot ``p`` holds rank ``p``'s."""
        assert mine.is_contiguous() and everyone.is_contiguous()
        n_bytes = mine.numel() * mine.element_size()
        assert everyone.numel() * everyone.element_size() == self.world * n_bytes
        _call("atx_all_gather", self._handle, _ptr(mine), _ptr(everyone), n_bytes, _stream())

    def exchange(self, send: list, recv: list) -> None:
        """``send[p]`` goes to rank ``p``, ``recv[p]`` is filled by rank ``p``; ``None`` / empty tensors skip the pair."""
        assert len(send) == len(recv) == self.world

        def table(tensors):
            ptrs = (c_void_p * self.world)()
            sizes = (c_int64 * self.world)()
            for p, t in enumerate(tensors):
                if t is not None and t.numel():
                    assert t.is_contiguous()
                    ptrs[p], sizes[p] = _ptr(t), t.numel() * t.element_size()
            return ptrs, sizes

        sp, sb = table(send)
        rp, rb = table(recv)
        _call("atx_exchange", self._handle, sp, sb, rp, rb, _stream())

    def gather_shards(self, full: torch.Tensor, byte_offsets: list[int]) -> None:
        """Every rank has filled its byte range ``[byte_offsets[rank], byte_offsets[rank + 1])`` of ``full``; afterwards all have all."""
        assert full.is_contiguous() and len(byte_offsets) == self.world + 1
        assert byte_offsets[-1] <= full.numel() * full.element_size()
        offs = (c_int64 * (self.world + 1))(*[int(o) for o in byte_offsets])
        _call("atx_gather_shards", self._handle, _ptr(full), offs, _stream())

    def destroy(self) -> None:
        if self._handle is not None:
            handle, self._handle = self._handle, None
            _call("atx_comm_destroy", handle)

    def __del__(self) -> None:  # best effort; explicit destroy() is the documented way
        try:
            self.destroy()
        except Exception:
            pass


def relayout(src, dst, *, n_pts, n_lev, src_pitch, dst_pitch, src_layout, dst_layout) -> None:
    assert src.dtype == dst.dtype
    _call(
        "atx_relayout", _ptr(src), _ptr(dst), n_pts, n_lev, src_pitch, dst_pitch, src_layout, dst_layout,
        dtype_code(src.dtype), _stream(),
    )
