"""Stack-level execution of per-point filters.

The reference maps a numpy statement over the selected fields in a Python loop
(R: filter.py:188-196).  Here the selected fields of a FieldList are gathered into
HBM stacks (``fields.group_into_stacks``) and each stack is transformed by ONE
``atx_pointwise_stack`` launch driven by a per-level program; unselected fields
pass through by identity, as in the reference.
"""

from __future__ import annotations

from typing import Any, Callable

import torch

from .. import native
from ..fields import FieldList, group_into_stacks, new_field_from_stack

# (op, use_mask, p0, p1)
LevelOp = tuple[int, int, float, float]


class PointMask:
    """A boolean mask over grid points, as bytes in HBM (1 = masked)."""

    def __init__(self, tensor: torch.Tensor, n_points: int) -> None:
        assert tensor.dtype == torch.uint8 and tensor.numel() >= n_points
        self.tensor = tensor
        self.n_points = n_points

    @classmethod
    def build(cls, values: torch.Tensor, n_points: int, *, cmp: int, threshold: float = 0.0, stride: int = 1) -> "PointMask":
        """``values CMP threshold`` evaluated on the device (R: apply_mask.py:160-163, remove_nans.py:101)."""
        tensor = torch.empty((n_points + 3) // 4 * 4 + 4, dtype=torch.uint8, device=values.device)
        native.mask_build(values, tensor, n=n_points, stride=stride, cmp=cmp, threshold=threshold)
        return cls(tensor, n_points)

    def window(self, lo: int, hi: int) -> "PointMask":
        """The mask of points ``[lo, hi)`` (a view): what a rank of a target-sharded job applies to its slice."""
        assert 0 <= lo <= hi <= self.n_points
        return PointMask(self.tensor[lo:], hi - lo)

    def for_fields(self, fields: list[Any], n_pts: int) -> "PointMask":
        """This mask as it applies to ``fields`` of ``n_pts`` points each: itself, or — for the slices a target-sharded
        regrid produced (``Field.target_range``) — its window; a mask of any other length is an error
        (R: apply_mask.py:185 would raise IndexError)."""
        if self.n_points == n_pts:
            return self
        ranges = {f.target_range() if hasattr(f, "target_range") else None for f in fields}
        if len(ranges) == 1:
            r = ranges.pop()
            if r is not None and r[2] == self.n_points and r[1] - r[0] == n_pts:
                return self.window(r[0], r[1])
        raise IndexError(f"boolean index did not match indexed array: mask has {self.n_points} points, field has {n_pts}")

    def count(self) -> int:
        return native.mask_count(self.tensor, self.n_points)

    def to_index(self) -> torch.Tensor:
        return native.mask_to_index(self.tensor, self.n_points)

    def numpy(self):
        return self.tensor[: self.n_points].cpu().numpy().astype(bool)


def run_level_ops(
    data: Any,
    select: Callable[[Any], bool],
    level_op: Callable[[Any], LevelOp],
    new_metadata: Callable[[Any], dict[str, Any]],
    point_mask: PointMask | None = None,
) -> FieldList:
    """Transform every field ``f`` with ``select(f)`` by ``level_op(f)``; returns a new FieldList.

    Output fields live in freshly allocated stacks (the input stacks are never
    written: fields of the input list stay valid, like the reference's new arrays).
    """
    fields = list(data)
    out = list(fields)
    positions = [i for i, f in enumerate(fields) if select(f)]
    if not positions:
        return FieldList(out)

    for group in group_into_stacks(fields, positions, sparse_ok=True):
        src = group.stack
        mask = None if point_mask is None else point_mask.for_fields(group.fields, src.n_pts)
        dst = src.new_like()
        stage: list[LevelOp] = [(native.OP_COPY, 0, 0.0, 0.0)] * src.n_lev  # levels outside the group: untouched
        for level, f in zip(group.levels, group.fields):
            stage[level] = level_op(f)
        prog = native.level_program([stage], src.device, cache=True)
        native.pointwise_stack(
            src.data, dst.data, n_pts=src.n_pts, n_lev=src.n_lev, x_pitch=src.pitch, y_pitch=dst.pitch,
            layout=src.layout, prog=prog, n_stage=1,
            point_mask=None if mask is None else mask.tensor,
        )
        for level, pos, f in zip(group.levels, group.positions, group.fields):
            out[pos] = new_field_from_stack(dst, level, template=f, metadata=new_metadata(f))
    return FieldList(out)
