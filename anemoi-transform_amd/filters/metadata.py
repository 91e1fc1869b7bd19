"""Filters that re-label or re-list fields without touching their values.

Mirrors of R: filters/fields/rename.py, clear_step.py, repeat_members.py, lambda_filters.py,
empty.py and icon_refinement_level.py.  They matter to the HBM path for
one reason: a recipe such as ``regrid | rename | rescale`` must not drag the data back to the
host in the middle.  Every filter here keeps a device field a device field (the new field
points at the same level of the same stack), and ``rename`` is fusable: inside a fused
pipeline it is a COPY stage that only changes metadata (``filters/fusion.py``).
"""

from __future__ import annotations

import datetime
import importlib
import re
from typing import Any, Callable

import numpy as np

from ..core import DispatchingFilter, Filter, SingleFieldFilter, filter_registry
from ..fields import (
    DerivedField,
    FieldList,
    new_empty_fieldlist,
    new_field_with_metadata,
    new_field_with_valid_datetime,
    new_fieldlist_from_list,
    to_datetime,
)
from ..gather import GatherPlan

_PLACEHOLDER = re.compile(r"{([\w:]+)}")


class FormatRename:
    """``key: "{param}_{levelist}"`` — the new value is a format of other metadata (R: rename.py:19-45).
    ``{name:d}`` asks for the value as a given eccodes type; ``:`` is not legal inside ``str.format``
    field names, so it travels as ``|``."""

    def __init__(self, what: str, fmt: str) -> None:
        self.what = what
        self.keys = _PLACEHOLDER.findall(fmt)
        self.format = _PLACEHOLDER.sub(lambda m: "{" + m.group(1).replace(":", "|") + "}", fmt)

    def new_value(self, field: Any) -> Any:
        values = [field.metadata(k) for k in self.keys]
        return self.format.format(**{k.replace(":", "|"): v for k, v in zip(self.keys, values)})

    def overrides(self, field: Any) -> dict[str, Any]:
        if field.metadata(self.what, default=None) is None:
            return {}
        return {self.what: self.new_value(field)}


class DictRename:
    """``key: {old: new, ...}`` (R: rename.py:48-64)."""

    def __init__(self, what: str, renaming: dict[Any, Any]) -> None:
        self.what = what
        self.renaming = renaming

    def overrides(self, field: Any) -> dict[str, Any]:
        current = field.metadata(self.what, default=None)
        if current is None or current not in self.renaming:
            return {}
        return {self.what: self.renaming[current]}


@filter_registry.register("rename_fields")
class Rename(SingleFieldFilter):
    """Rename metadata values, by table or by format string (R: rename.py:67-135).  Renamers apply in
    configuration order and a later one sees the result of an earlier one."""

    def prepare_filter(self) -> None:
        renamers = []
        for key, value in self.config.items():
            if isinstance(value, str):
                renamers.append(FormatRename(key, value))
            elif isinstance(value, dict):
                renamers.append(DictRename(key, value))
            else:
                raise ValueError(f"Invalid value for rename: {key}: {value}")
        self.renamers = tuple(renamers)

    def forward_transform(self, field: Any) -> Any:
        for renamer in self.renamers:
            changed = renamer.overrides(field)
            if changed:
                field = new_field_with_metadata(template=field, **changed)
        return field

    def fused_metadata(self, field: Any) -> dict[str, Any]:
        """All overrides of this filter for ``field`` at once (what a fused COPY stage attaches)."""
        out: dict[str, Any] = {}
        for renamer in self.renamers:
            changed = renamer.overrides(field)
            if changed:
                out.update(changed)
                field = DerivedField(field, metadata=changed)
        return out


class RenameDispatcher(DispatchingFilter):
    """Top-level ``rename`` (R: filters/rename.py:19-33).  ``columns=`` selects the tabular filter, which is
    outside this package's scope (observations)."""

    def __init__(self, **config: Any) -> None:
        if set(config.keys()) == {"columns"}:
            raise NotImplementedError("rename(columns=...) is the tabular (DataFrame) filter; this package covers gridded fields")
        self.filter = Rename(**config)

    def forward_fields(self, data: Any) -> Any:
        return self.filter.forward(data)


filter_registry.register("rename", RenameDispatcher)


@filter_registry.register("clear_step")
class ClearStepFilter(Filter):
    """``valid_datetime -= step`` hours (R: clear_step.py:24-51): the field is relabelled with its base time."""

    def __init__(self) -> None:
        super().__init__()

    def forward(self, data: Any) -> FieldList:
        out = []
        for field in data:
            valid = to_datetime(field.metadata("valid_datetime"))
            step = field.metadata("step")
            out.append(new_field_with_valid_datetime(field, valid - datetime.timedelta(hours=step)))
        return new_fieldlist_from_list(out)


def _as_int_list(value: Any) -> list[int]:
    """``[1, 3, 5]``, ``"1/3/5"``, ``"1/to/5"``, ``"1/to/9/by/2"`` or a single int (anemoi-utils ``make_list_int``)."""
    if isinstance(value, (list, tuple)):
        return [int(v) for v in value]
    if isinstance(value, (int, np.integer)):
        return [int(value)]
    if isinstance(value, str):
        bits = value.split("/")
        if len(bits) == 3 and bits[1].lower() == "to":
            return list(range(int(bits[0]), int(bits[2]) + 1))
        if len(bits) == 5 and bits[1].lower() == "to" and bits[3].lower() == "by":
            return list(range(int(bits[0]), int(bits[2]) + 1, int(bits[4])))
        return [int(b) for b in bits]
    raise ValueError(f"Cannot make list of int from {value!r}")


@filter_registry.register("repeat_members")
class RepeatMembers(Filter):
    """Every field is listed once per ensemble member with ``number = member + 1`` (R: repeat_members.py:22-125).
    The copies share the field's data: on the device they are the same level of the same stack."""

    def __init__(self, *, numbers: Any = None, members: Any = None, count: int | None = None) -> None:
        if sum(x is not None for x in (members, count, numbers)) != 1:
            raise ValueError("Exactly one of members, count or numbers must be given")
        if numbers is not None:
            members = [n - 1 for n in _as_int_list(numbers)]
        if count is not None:
            members = list(range(count))
        self.members = _as_int_list(members)

    def forward(self, data: Any) -> FieldList:
        out = []
        for field in data:
            for member in self.members:
                out.append(new_field_with_metadata(field, number=member + 1))
        return new_fieldlist_from_list(out)


@filter_registry.register("earthkitfieldlambda")
class FieldLambdaFilter(SingleFieldFilter):
    """Apply a user function ``fn(field, *fn_args, **fn_kwargs) -> field`` to the selected fields
    (R: lambda_filters.py:18-131).  The function sees this package's fields: ``to_numpy()`` for the host
    copy, ``stack_ref()`` for the HBM level."""

    required_inputs = ("fn", "param")
    optional_inputs = {"fn_args": None, "fn_kwargs": None, "backward_fn": None}

    def prepare_filter(self) -> None:
        cfg = self._config
        cfg["fn_args"] = [] if cfg["fn_args"] is None else cfg["fn_args"]
        cfg["fn_kwargs"] = {} if cfg["fn_kwargs"] is None else cfg["fn_kwargs"]
        if not isinstance(cfg["fn_args"], list):
            raise ValueError(f"Expected 'fn_args' to be a list. Got {cfg['fn_args']} instead.")
        if not isinstance(cfg["fn_kwargs"], dict):
            raise ValueError(f"Expected 'fn_kwargs' to be a dictionary. Got {cfg['fn_kwargs']} instead.")
        if not isinstance(cfg["fn"], str):
            raise ValueError(f"Expected 'fn' to be a string. Got {cfg['fn']} instead.")
        # the reference rejects a missing backward_fn here (lambda_filters.py:75-76) although it documents it as optional
        # and guards for None in backward_transform (:91-92); the documented behaviour is kept
        if cfg["backward_fn"] is not None and not isinstance(cfg["backward_fn"], str):
            raise ValueError(f"Expected 'backward_fn' to be a string. Got {cfg['backward_fn']} instead.")
        cfg["fn"] = self._import_fn(cfg["fn"])
        if cfg["backward_fn"] is not None:
            cfg["backward_fn"] = self._import_fn(cfg["backward_fn"])

    def forward_select(self) -> dict[str, Any]:
        return {"param": self.param}

    def forward_transform(self, field: Any) -> Any:
        return self.fn(field, *self.fn_args, **self.fn_kwargs)

    def backward_transform(self, field: Any) -> Any:
        if self.backward_fn is None:
            raise ValueError("Backward function is undefined.")
        return self.backward_fn(field, *self.fn_args, **self.fn_kwargs)

    @staticmethod
    def _import_fn(path: str) -> Callable[..., Any]:
        try:
            module_name, fn_name = path.rsplit(".", 1)
            return getattr(importlib.import_module(module_name), fn_name)
        except Exception as e:
            raise ValueError(f"Could not import function {path}") from e

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(fn={self.fn},backward_fn={self.backward_fn},param={self.param},"
                f"fn_args={self.fn_args},fn_kwargs={self.fn_kwargs},)")


@filter_registry.register("empty")
class Empty(Filter):
    """Swallows its input (debugging aid, R: empty.py:17-33)."""

    def __init__(self) -> None:
        super().__init__()

    def forward(self, data: Any) -> FieldList:
        return new_empty_fieldlist()


@filter_registry.register("icon_refinement_level")
class IconRefinement(Filter):
    """Nearest-neighbour interpolation to the cells of an ICON grid up to a refinement level
    (R: icon_refinement_level.py:24-85): one k = 1 gather launch per stack.

    ``grid`` is a ``.npz`` with ``clat`` / ``clon`` (radians, as in ICON grid files) or ``latitudes`` /
    ``longitudes`` (degrees) and, when ``refinement_level_c`` is given, ``refinement_level_c`` per cell —
    the three variables R: grids/icon.py:22-53 reads from the netCDF grid file (xarray / netCDF are not
    available here, so the npz carries them), or a dict with the same keys.
    """

    def __init__(self, *, grid: Any, refinement_level_c: int | None) -> None:
        self.grid = grid
        self.refinement_level_c = refinement_level_c
        self.latitudes, self.longitudes = self._read_grid(grid, refinement_level_c)
        self.nearest_grid_points = None
        self._plan: GatherPlan | None = None

    @staticmethod
    def _read_grid(grid: Any, level: int | None) -> tuple[np.ndarray, np.ndarray]:
        ds = dict(np.load(grid)) if isinstance(grid, str) else dict(grid)
        keep = slice(None) if level is None else np.asarray(ds["refinement_level_c"]) <= level
        if "clat" in ds:
            return np.rad2deg(np.asarray(ds["clat"])[keep]), np.rad2deg(np.asarray(ds["clon"])[keep])
        return np.asarray(ds["latitudes"])[keep], np.asarray(ds["longitudes"])[keep]

    def forward(self, fields: Any) -> FieldList:
        from ..fields import group_into_stacks, new_field_from_stack
        from ..interp import nearest_grid_points

        fields = list(fields)
        if not fields:
            return new_empty_fieldlist()
        if self._plan is None:  # all fields are assumed to share the first one's grid (R: :62-72)
            latitudes, longitudes = fields[0].grid_points()
            self.nearest_grid_points = nearest_grid_points(latitudes, longitudes, self.latitudes, self.longitudes)
            self._plan = GatherPlan(len(latitudes), len(self.nearest_grid_points), index=self.nearest_grid_points)
        out: list[Any] = [None] * len(fields)
        for group in group_into_stacks(fields):
            regridded = self._plan.apply(group.stack)
            for level, (pos, f) in enumerate(zip(group.positions, group.fields)):
                new = new_field_from_stack(regridded, level, template=f, latitudes=self.latitudes, longitudes=self.longitudes)
                new.resolution = f"mrl{self.refinement_level_c}"
                out[pos] = new
        return new_fieldlist_from_list(out)
