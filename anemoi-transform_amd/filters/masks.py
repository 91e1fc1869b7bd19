"""Spatial mask filters on HBM stacks.

  apply_mask_fields   R: filters/fields/apply_mask.py:39-245
  remove_nans_fields  R: filters/fields/remove_nans.py:25-119

Masks are built on the device (``atx_mask_build``), applied by the per-level
program evaluator (``use_mask``), and ``remove_nans``' boolean compress becomes a
stable compaction (``atx_mask_to_index``) followed by the k = 1 gather kernel.
"""

from __future__ import annotations

from typing import Any

import numpy as np
import torch

from .. import native
from ..core import Filter, filter_registry
from ..fields import (
    FieldList,
    FieldSelection,
    group_into_stacks,
    new_field_from_stack,
)
from ..gather import GatherPlan
from .. import stack as _stack
from ..stack import COLUMNS
from .engine import PointMask, run_level_ops
from .pointwise import load_mask_file

# R: apply_mask.py:23-36 — 12 spellings, 6 comparisons
OPERATORS = {
    ">": native.CMP_GT, "<": native.CMP_LT, "==": native.CMP_EQ, "!=": native.CMP_NE, ">=": native.CMP_GE, "<=": native.CMP_LE,
    "gt": native.CMP_GT, "lt": native.CMP_LT, "eq": native.CMP_EQ, "ne": native.CMP_NE, "ge": native.CMP_GE, "le": native.CMP_LE,
}


def _level_tensor(field: Any) -> tuple[torch.Tensor, int, int]:
    """Device view of one field: ``(tensor, stride, n_points)``; uploads a host field."""
    ref = field.stack_ref() if hasattr(field, "stack_ref") else None
    if ref is not None:
        stack, level = ref
        view = stack.level_view(level)
        return view, (stack.pitch if stack.layout == COLUMNS else 1), stack.n_pts
    host = np.ascontiguousarray(field.to_numpy(flatten=True))
    if host.dtype not in (np.float32, np.float64):
        host = host.astype(np.float64)
    return torch.from_numpy(host).to(_stack.device()), 1, host.size


@filter_registry.register("apply_mask_fields")
class MaskVariable(Filter):
    """Set fields to NaN where a mask field equals ``mask_value`` or meets a ``threshold`` condition.

    The mask comes from a file (``path``) or from a field of the stream
    (``mask_param``; consumed unless ``return_mask``).  ``param`` restricts the
    masking to some variables; ``rename`` appends ``_{rename}`` to the parameter name.
    """

    def __init__(
        self,
        *,
        path: str | None = None,
        mask_param: str | None = None,
        mask_value: float | None = None,
        threshold: float | None = None,
        threshold_operator: str = ">",
        rename: str | None = None,
        param: str | list[str] | None = None,
        return_mask: bool = False,
    ) -> None:
        # the configuration is kept as plain attributes (the reference's filters expose their config that way, R: filter.py:184-186)
        vars(self).update(path=path, mask_param=mask_param, mask_value=mask_value, threshold=threshold,
                          threshold_operator=threshold_operator, rename=rename, return_mask=return_mask)
        self.param = [param] if isinstance(param, str) else param
        self.mask: PointMask | None = None
        self._file_values: np.ndarray | None = None
        self.prepare_filter()
        self._forward_selection = FieldSelection(**self.forward_select())

    def prepare_filter(self) -> None:
        # R: apply_mask.py:140-158 — same checks, same messages
        if (self.path is None) == (self.mask_param is None):
            raise ValueError("Exactly one of `path` or `mask_param` must be provided.")
        if (self.mask_value is None) == (self.threshold is None):
            raise ValueError("Exactly one of `mask_value` or `threshold` must be provided.")
        if self.threshold is not None and self.threshold_operator not in OPERATORS:
            raise ValueError(
                f"Invalid threshold operator: {self.threshold_operator}. Valid operators are: {', '.join(OPERATORS.keys())}."
            )
        if self.path is not None:
            self._file_values = np.asarray(load_mask_file(self.path)).reshape(-1)

    def _compute_mask(self, values: torch.Tensor, stride: int, n_points: int) -> PointMask:
        # R: apply_mask.py:160-163
        if self.threshold is not None:
            return PointMask.build(values, n_points, cmp=OPERATORS[self.threshold_operator], threshold=self.threshold, stride=stride)
        return PointMask.build(values, n_points, cmp=native.CMP_EQ, threshold=self.mask_value, stride=stride)

    def forward_select(self) -> dict[str, Any]:
        return {"param": self.param} if self.param is not None else {}

    def _new_metadata(self, field: Any) -> dict[str, Any]:
        if self.rename is None:
            return {}
        return {"param": f"{field.metadata('param')}_{self.rename}"}  # R: apply_mask.py:187-190

    def _separate_mask_and_fields(self, fields: Any) -> tuple[PointMask, list[Any]]:
        # R: apply_mask.py:194-218
        if self.mask_param is None:
            if self.mask is None:
                host = self._file_values
                if host.dtype not in (np.float32, np.float64):
                    host = host.astype(np.float64)
                self.mask = self._compute_mask(torch.from_numpy(np.ascontiguousarray(host)).to(_stack.device()), 1, host.size)
            return self.mask, list(fields)
        # the mask is the FIRST field carrying `mask_param`; every field of that name leaves the stream unless `return_mask`
        fields = list(fields)
        is_mask = [f.metadata("param") == self.mask_param for f in fields]
        if not any(is_mask):
            raise ValueError(f"Mask parameter '{self.mask_param}' not found in input data.")
        mask_field = fields[is_mask.index(True)]
        remaining = fields if self.return_mask else [f for f, m in zip(fields, is_mask) if not m]
        return self._compute_mask(*_level_tensor(mask_field)), remaining

    def forward_transform(self, field: Any) -> Any:
        return run_level_ops([field], lambda f: True, lambda f: (native.OP_COPY, 1, 0.0, 0.0), self._new_metadata, self.mask)[0]

    def forward(self, fields: Any) -> FieldList:
        self.mask, remaining = self._separate_mask_and_fields(fields)
        return run_level_ops(
            remaining, self._forward_selection.match, lambda f: (native.OP_COPY, 1, 0.0, 0.0), self._new_metadata, self.mask
        )


@filter_registry.register("remove_nans_fields")
class RemoveNaNs(Filter):
    """Drop, from every field, the grid points where the first field (or the first field of
    ``param``) is NaN; latitudes / longitudes shrink accordingly.  The mask is computed
    once and cached (R: remove_nans.py:90-105)."""

    def __init__(self, *, method: str = "mask", check: bool = False, param: str | None = None):
        self.method = method
        self.check = check
        self.param = param
        assert method == "mask", f"Method {method} not implemented"
        assert not check, "Check not implemented"
        self._mask = None  # host bool mask, as in the reference
        self._plan: GatherPlan | None = None
        self._latitudes = None
        self._longitudes = None

    def _prepare(self, fields: Any) -> None:
        if self.param is None:
            first = fields[0]
        else:
            for first in fields:
                if first.metadata("param") == self.param:
                    break
            else:
                raise ValueError(f"{self.param=} not found in\n{getattr(fields, 'ls', fields)}")
        values, stride, n_points = _level_tensor(first)
        keep = PointMask.build(values, n_points, cmp=native.CMP_NOTNAN, stride=stride)  # ~isnan
        index = keep.to_index()  # ascending: numpy boolean indexing order
        self._mask = keep.numpy()
        self._plan = GatherPlan(n_points, index.numel(), index=index.cpu().numpy())
        latitudes, longitudes = first.grid_points()
        self._latitudes = latitudes[self._mask]
        self._longitudes = longitudes[self._mask]

    def forward(self, fields: Any) -> FieldList:
        fields = fields if isinstance(fields, FieldList) else FieldList(list(fields))
        if self._plan is None:
            self._prepare(fields)
        out: list[Any] = [None] * len(fields)
        for group in group_into_stacks(fields):
            if group.stack.n_pts != self._plan.n_src:
                # R: remove_nans.py:113 `data[self._mask]` — numpy's own refusal of a field on another grid than the first one
                raise IndexError(f"boolean index did not match indexed array along axis 0; size of axis is {group.stack.n_pts} "
                                 f"but size of corresponding boolean axis is {self._plan.n_src}")
            compressed = self._plan.apply(group.stack)
            for level, (pos, f) in enumerate(zip(group.positions, group.fields)):
                out[pos] = new_field_from_stack(
                    compressed, level, template=f, latitudes=self._latitudes, longitudes=self._longitudes
                )
        return FieldList(out)
