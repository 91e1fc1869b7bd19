"""Pipeline fusion: chained filters as ONE launch per stack.

The reference materialises a full FieldList between every two filters of a pipeline
(R: workflows/pipeline.py:46-48): regrid -> orog_to_z -> convert on an O2560 stack is
three passes over HBM-sized data.  Here a run of per-point filters is folded into
a multi-stage per-level program, and when the run follows a ``regrid`` the program
is evaluated in the gather kernel's epilogue (``atx_regrid_ell`` / ``atx_regrid_csr``
``prog`` argument), so the interpolated value is transformed in registers and
stored once.  Results are identical to running the filters one after the other
(same operators, same order, same rounding) — checked in tests/test_fusion.py.

What can be fused: filters whose effect on a field is a function of the field's
METADATA only — the ``StackFieldFilter`` family (rescale, convert, orog_to_z, clip,
impute_nans, lnsp_to_sp, glacier_mask), ``apply_mask`` with a file mask, their
dispatchers and ``ReversedTransform`` wrappers.  ``apply_mask`` with ``mask_param``
(mask taken from a field of the stream), ``remove_nans`` and user-defined filters
end a fused segment and run as usual.
"""

from __future__ import annotations

import os
from typing import Any, Callable

from .. import native
from ..core import Pipeline, ReversedTransform
from ..fields import DerivedField, FieldList, group_into_stacks, new_field_from_stack
from .engine import LevelOp, PointMask
from .masks import MaskVariable
from .metadata import Rename
from .pointwise import StackFieldFilter
from .regrid import RegridFilter

MAX_STAGES = 8  # atx.h: n_stage <= 8
COPY: LevelOp = (native.OP_COPY, 0, 0.0, 0.0)


class Stage:
    """One fusable filter, reduced to metadata-level callables."""

    def __init__(self, select: Callable[[Any], bool], level_op: Callable[[Any], LevelOp],
                 new_metadata: Callable[[Any], dict[str, Any]], mask: Callable[[], PointMask | None]) -> None:
        self.select = select
        self.level_op = level_op
        self.new_metadata = new_metadata
        self.mask = mask


def as_stage(f: Any, backward: bool = False) -> Stage | None:
    """The fusable form of a filter, or ``None``."""
    if isinstance(f, ReversedTransform):
        return as_stage(f.filter, not backward)
    if isinstance(f, StackFieldFilter):
        if backward:
            if type(f).backward_level_op is StackFieldFilter.backward_level_op:
                return None  # not reversible: let the ordinary path raise
            return Stage(f._backward_selection.match, f.backward_level_op, f.backward_metadata, f.point_mask)
        return Stage(f._forward_selection.match, f.forward_level_op, f.forward_metadata, f.point_mask)
    if isinstance(f, Rename):
        if backward:
            return None
        # metadata only: a COPY stage, so a rename between two per-point filters does not end the fused run
        return Stage(lambda field: bool(f.fused_metadata(field)), lambda field: COPY, f.fused_metadata, lambda: None)
    if isinstance(f, MaskVariable):
        if backward or f.mask_param is not None:
            return None

        def file_mask() -> PointMask:
            return f._separate_mask_and_fields([])[0]

        return Stage(f._forward_selection.match, lambda field: (native.OP_COPY, 1, 0.0, 0.0), f._new_metadata, file_mask)
    # dispatchers of filters/__init__.py keep their field filter in one of these attributes
    for attr in ("field_filter", "filter"):
        inner = f.__dict__.get(attr) if hasattr(f, "__dict__") else None
        if inner is not None and inner is not f and not isinstance(f, (RegridFilter, Pipeline)):
            return as_stage(inner, backward)
    return None


def flatten(filters: list[Any]) -> list[Any]:
    """``a | b | c`` nests two-element pipelines (R: transform.py:116-131); undo that."""
    out = []
    for f in filters:
        if type(f) is Pipeline:
            out.extend(flatten(f.filters))
        else:
            out.append(f)
    return out


def _gather_head(f: Any):
    """The interpolator of a plain forward ``regrid`` filter, if it runs a GatherPlan."""
    if isinstance(f, RegridFilter) and hasattr(f.interpolator, "plan_for"):
        return f
    return None


def _run_segment(data: Any, head: RegridFilter | None, stages: list[Stage]) -> FieldList:
    fields = list(data)
    n = len(fields)
    # follow the metadata of every field through the stages (selection of a later stage sees
    # the renames of an earlier one), recording one operator per (stage, field)
    proxies = list(fields)
    ops: list[list[LevelOp]] = []
    touched = [False] * n
    mask: PointMask | None = None
    for st in stages:
        row = []
        for i in range(n):
            if st.select(proxies[i]):
                op = st.level_op(proxies[i])
                relabel = st.new_metadata(proxies[i])
                if relabel:  # a stage that changes no metadata needs no new view of the field
                    proxies[i] = DerivedField(proxies[i], metadata=relabel)
                touched[i] = touched[i] or op is not COPY  # a pure relabelling needs no launch
                if op[1]:
                    m = st.mask()
                    assert mask is None or mask is m, "one point mask per fused launch"
                    mask = m
            else:
                op = COPY
            row.append(op)
        ops.append(row)

    out: list[Any] = list(proxies)  # untouched fields: the original, or its relabelled view (same data, same stack level)
    if head is None:
        positions = [i for i in range(n) if touched[i]]
        if not positions:
            return FieldList(out)
        for group in group_into_stacks(fields, positions, sparse_ok=True):
            src = group.stack
            # the point mask belongs to the fields some stage masks; a group none of whose operators uses it (another grid,
            # say) must not be measured against it — filter by filter those fields never meet the mask either
            needs_mask = mask is not None and any(row[p][1] for row in ops for p in group.positions)
            group_mask = mask.for_fields(group.fields, src.n_pts) if needs_mask else None
            dst = src.new_like()
            stages = []
            for row in ops:
                stage = [COPY] * src.n_lev
                for level, p in zip(group.levels, group.positions):
                    stage[level] = row[p]
                stages.append(stage)
            prog = native.level_program(stages, src.device, cache=True)
            native.pointwise_stack(src.data, dst.data, n_pts=src.n_pts, n_lev=src.n_lev, x_pitch=src.pitch, y_pitch=dst.pitch,
                                   layout=src.layout, prog=prog, n_stage=len(ops),
                                   point_mask=None if group_mask is None else group_mask.tensor)
            for level, pos in zip(group.levels, group.positions):
                out[pos] = new_field_from_stack(dst, level, template=proxies[pos])
        return FieldList(out)

    interp = head.interpolator
    for group in group_into_stacks(fields):
        plan = interp.plan_for(group.fields[0])
        lat, lon = interp.out_latlon(group.fields[0])
        window = None
        full_targets = plan.n_tgt
        if head.shard is not None:
            lo, hi = plan.shard_range(*head.shard)
            window = (lo, hi, full_targets)
            plan, lat, lon = interp._sharded(plan, head.shard), lat[lo:hi], lon[lo:hi]
        kwargs = {}
        if any(touched[p] for p in group.positions):
            tgt_mask = mask if any(row[p][1] for row in ops for p in group.positions) else None  # only groups that use it
            if tgt_mask is not None and mask.n_points != plan.n_tgt:
                if window is None or mask.n_points != full_targets:
                    raise IndexError(f"boolean index did not match indexed array: mask has {mask.n_points} points, field has {plan.n_tgt}")
                tgt_mask = mask.window(window[0], window[1])  # a full-grid mask on this rank's slice of the targets
            kwargs = dict(prog=native.level_program([[row[p] for p in group.positions] for row in ops], group.stack.device, cache=True),
                          n_stage=len(ops), tgt_mask=None if tgt_mask is None else tgt_mask.tensor)
        regridded = plan.apply(group.stack, **kwargs)
        for level, pos in enumerate(group.positions):
            out[pos] = new_field_from_stack(regridded, level, template=proxies[pos], latitudes=lat, longitudes=lon, target_range=window)
    return FieldList(out)


def forward_fused(filters: list[Any], data: Any) -> Any:
    """``Pipeline.forward`` with fusable runs collapsed into single launches."""
    if os.environ.get("ATX_NO_FUSION"):
        for f in filters:
            data = f.forward(data)
        return data
    flat = flatten(filters)
    i = 0
    while i < len(flat):
        head = _gather_head(flat[i])
        j = i + 1 if head is not None else i
        stages: list[Stage] = []
        masks_seen = 0
        while j < len(flat) and len(stages) < MAX_STAGES:
            st = as_stage(flat[j])
            if st is None:
                break
            uses_mask = isinstance(flat[j], MaskVariable) or (st.mask() is not None)
            if uses_mask and masks_seen:
                break  # a second point mask needs its own launch
            masks_seen += int(uses_mask)
            stages.append(st)
            j += 1
        if head is not None and stages:
            data = _run_segment(data, head, stages)
        elif head is None and len(stages) >= 2:
            data = _run_segment(data, None, stages)
        else:
            data = flat[i].forward(data)
            j = i + 1
        i = j
    return data
