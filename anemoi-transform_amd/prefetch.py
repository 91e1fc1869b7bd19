"""Host-fed jobs: put the NEXT FieldList into HBM while the current one is being transformed.

The reference processes one FieldList after the other on the host (R: workflows/pipeline.py:46-48: ``for f in filters: data =
f.forward(data)``); its unit of work per date is ~100 GRIB fields decoded into host arrays.  Here a host-fed ``forward`` is
PCIe-bound, not kernel-bound: 137 O1280 float32 fields take ~70 ms to stage and upload (threads filling pinned chunks while the
DMA of the previous chunk runs, ``stack._upload_rows``), 0.5 ms to regrid and ~20 ms to bring back and hand out as arrays.  The
upload depends on nothing but the input, so it can run one FieldList ahead on a background thread:

    for fields in prefetch_to_device(source_of_fieldlists):      # device-backed FieldLists, in order
        out = pipeline.forward(fields)                            # launches only: the stacks are already in HBM
        ...                                                       # to_numpy / writing, while the next upload is under way

Nothing else changes: the fields a filter sees are ordinary ``Field`` objects whose values live in an HBM stack (``to_device``),
with the metadata, grid and order of the originals.
"""

from __future__ import annotations

import queue
import threading
from typing import Any, Iterable, Iterator

from .fields import FieldList, group_into_stacks, new_field_from_stack

__all__ = ["to_device", "prefetch_to_device"]


def to_device(fields: Iterable[Any]) -> FieldList:
    """The same fields, in the same order, backed by HBM stacks (host fields of one grid go up as ONE stack; fields that already
    live on the device are kept as they are)."""
    fields = list(fields)
    out: list[Any] = [None] * len(fields)
    for group in group_into_stacks(fields, sparse_ok=True):
        for j, (pos, field) in enumerate(zip(group.positions, group.fields)):
            level = j if group.levels is None else group.levels[j]
            ref = field.stack_ref() if hasattr(field, "stack_ref") else None
            out[pos] = field if (ref is not None and ref[0] is group.stack) else new_field_from_stack(group.stack, level, template=field)
    return FieldList(out)


_DONE = object()


class _Uploaded:
    """A FieldList whose stacks were filled on the producer's stream, and the event that says when."""

    def __init__(self, fields: FieldList, done) -> None:
        self.fields, self.done = fields, done

    def hand_over(self) -> FieldList:
        """Called on the consumer's thread: its current stream waits (on the device) for the upload, and the caching allocator learns
        that the stacks — allocated on the producer's stream — are now used on this one."""
        if self.done is not None:
            import torch

            current = torch.cuda.current_stream()
            current.wait_event(self.done)
            seen = set()
            for field in self.fields:
                ref = field.stack_ref() if hasattr(field, "stack_ref") else None
                if ref is not None and id(ref[0]) not in seen and ref[0].data.is_cuda:
                    seen.add(id(ref[0]))
                    ref[0].data.record_stream(current)
        return self.fields


def prefetch_to_device(fieldlists: Iterable[Iterable[Any]], depth: int = 1) -> Iterator[FieldList]:
    """Yield ``to_device(fl)`` for every ``fl`` of ``fieldlists``, in order, with up to ``depth`` uploads running ahead of the
    consumer on a background thread.  An exception raised while reading or uploading an item is re-raised at the point where that
    item would have been yielded; closing the generator early stops the thread after the upload in progress."""
    if depth < 1:
        raise ValueError("depth must be at least 1")
    ready: "queue.Queue[Any]" = queue.Queue(maxsize=depth)
    stop = threading.Event()

    def producer() -> None:
        import contextlib

        import torch

        side = None
        if torch.cuda.is_available():
            from . import stack as _stack

            dev = _stack.device()
            if dev.type == "cuda":
                torch.cuda.set_device(dev)
                # the producer's OWN stream: its uploads (and the layout kernel behind them) neither queue behind the consumer's
                # launches on the default stream nor put wait markers into it — the consumer gets an event to wait on instead
                side = torch.cuda.Stream(dev)
        try:
            for fl in fieldlists:
                if stop.is_set():
                    return
                with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                    item = to_device(fl)
                    done = None
                    if side is not None:
                        done = torch.cuda.Event()
                        done.record(side)
                item = _Uploaded(item, done)
                while not stop.is_set():
                    try:
                        ready.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        continue
        except BaseException as e:  # handed to the consumer
            ready.put(e)
            return
        ready.put(_DONE)

    thread = threading.Thread(target=producer, name="atx-prefetch", daemon=True)
    thread.start()
    try:
        while True:
            item = ready.get()
            if item is _DONE:
                return
            if isinstance(item, BaseException):
                raise item
            yield item.hand_over()
    finally:
        stop.set()
        while True:  # unblock a producer waiting on a full queue
            try:
                ready.get_nowait()
            except queue.Empty:
                break
        thread.join(timeout=60)
