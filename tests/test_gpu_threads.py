"""The threading contract of the C ABI (include/atx.h "Threading"; SURVEY.md §8b: "functions are re-entrant, enqueue on the passed
hipStream_t and return without sync; one host thread per device or one thread driving 8 streams"): several host threads, each on its
own HIP stream and its own buffers, call the library AT THE SAME TIME — ctypes releases the GIL for the duration of every call —
and every result is held to the CPU oracle.  The state the header says is per thread (the tuning hook, the error string) must not leak
from one thread into another."""

from __future__ import annotations

import threading

import numpy as np
import pytest
import torch

from anemoi_transform_amd import native
from anemoi_transform_amd.stack import COLUMNS, Stack
from oracle import oracle

pytestmark = pytest.mark.gpu

N_THREADS = 4
ROUNDS = 6


def worker(tid: int, dev, barrier: threading.Barrier, errors: list, notes: dict) -> None:
    try:
        rng = np.random.default_rng(1000 + tid)
        np_dtype, k = (np.float64, 4) if tid % 2 == 0 else (np.float32, 3)
        n_src, n_tgt, n_lev = 20_000 + 1_111 * tid, 30_000 + 777 * tid, 5 + 8 * tid  # every thread its own shapes (137-ish columns on thread 3 would be 29)
        tiled = tid in (1, 3)  # atx_set_tuning is per calling thread: these two run the TILED regrid kernels, the others the direct one
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            if tiled:
                native.set_tuning(16)
            fields = (280.0 + 30.0 * rng.standard_normal((n_lev, n_src))).astype(np_dtype)
            fields[rng.random(fields.shape) < 0.02] = np.nan
            idx = rng.integers(0, n_src, size=(n_tgt, k)).astype(np.int32)
            w = rng.random((n_tgt, k))
            w = (w / w.sum(axis=1, keepdims=True)).astype(np_dtype)
            indptr = (np.arange(n_tgt + 1, dtype=np.int64) * k).astype(np.int32)
            want_regrid = np.stack([oracle.csr_apply(w.reshape(-1), idx.reshape(-1), indptr, (n_tgt, n_src), f) for f in fields])
            scale, offset = np_dtype(1.0 + 0.25 * tid), np_dtype(-273.15)
            want_point = oracle.rescale_forward(fields, scale, offset)
            clean = np.where(np.isnan(fields), np_dtype(280.0), fields)
            want_min, want_max = clean.min(), clean.max()
            mask_field = fields[0]
            want_index = np.flatnonzero(oracle.not_nan_mask(mask_field)).astype(np.int32)

            src = Stack.from_fields(fields, dev=dev)
            clean_d = Stack.from_fields(clean, dev=dev)
            idx_d, w_d = torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev)
            m_d = torch.from_numpy(mask_field).to(dev)
            prog = native.level_program([[(native.OP_AFFINE, 0, float(scale), float(offset))] * n_lev], dev)
            stream.synchronize()
            barrier.wait(timeout=120)  # all threads enter the loop together
            for it in range(ROUNDS):
                out = Stack.empty(n_tgt, n_lev, src.data.dtype, dev, COLUMNS)
                native.regrid_ell(src.data, out.data, idx_d, w_d, n_src=n_src, n_tgt=n_tgt, k=k, n_lev=n_lev, src_pitch=src.pitch,
                                  out_pitch=out.pitch, layout=COLUMNS)
                y = src.new_like()
                native.pointwise_stack(src.data, y.data, n_pts=n_src, n_lev=n_lev, x_pitch=src.pitch, y_pitch=y.pitch, layout=COLUMNS, prog=prog,
                                       n_stage=1)
                lo, hi = native.reduce_stack(clean_d.data, native.RED_MINMAX, n_pts=n_src, n_lev=n_lev, pitch=clean_d.pitch, layout=COLUMNS)
                mask = torch.empty(n_src, dtype=torch.uint8, device=dev)
                native.mask_build(m_d, mask, n=n_src, cmp=native.CMP_NOTNAN)
                index = native.mask_to_index(mask)
                if tid == 2:  # this thread also makes the library refuse a call, every round: k = 0
                    with pytest.raises(ValueError):
                        native.regrid_ell(src.data, out.data, idx_d, w_d, n_src=n_src, n_tgt=n_tgt, k=0, n_lev=n_lev, src_pitch=src.pitch,
                                          out_pitch=out.pitch, layout=COLUMNS)
                stream.synchronize()
                assert np.array_equal(out.numpy(), want_regrid, equal_nan=True), f"thread {tid} round {it}: regrid"
                assert np.array_equal(y.numpy(), want_point, equal_nan=True), f"thread {tid} round {it}: pointwise"
                assert (lo, hi) == (float(want_min), float(want_max)), f"thread {tid} round {it}: reduce"
                assert np.array_equal(index.cpu().numpy(), want_index), f"thread {tid} round {it}: mask_to_index"
            notes[tid] = native.load().atx_last_error().decode()
            if tiled:
                native.set_tuning(0)
    except BaseException as e:  # noqa: BLE001 - reported by the main thread
        errors.append((tid, repr(e)))
        try:
            barrier.abort()
        except Exception:  # noqa: BLE001
            pass


def test_four_threads_on_their_own_streams(dev):
    """4 host threads x 6 rounds of regrid (direct on two threads, tiled on the two whose thread-local tuning asks for it, float64 and
    float32), a per-point program, a one-pass min+max and a mask compaction, each thread on its own stream and buffers: every result
    equals the oracle's, the thread that provokes ATX_EINVAL sees its message, the other three never see one."""
    native.load()
    assert native.load().atx_last_error() is not None
    barrier = threading.Barrier(N_THREADS)
    errors: list = []
    notes: dict = {}
    threads = [threading.Thread(target=worker, args=(t, dev, barrier, errors, notes), name=f"atx-{t}") for t in range(N_THREADS)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a worker thread did not finish"
    assert not errors, errors
    assert set(notes) == set(range(N_THREADS))
    assert "k" in notes[2] and notes[2] != ""  # the refused call's message, in the thread that made the call
    assert all(notes[t] == "" for t in (0, 1, 3)), notes  # ... and in no other: the error string is per thread
    torch.cuda.synchronize()
