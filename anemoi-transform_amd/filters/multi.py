"""Multi-input per-point filters: fields matched by metadata, computed for all groups in one launch.

Mirror of R: filters/fields/matching.py (``MatchingSpec``, ``MatchingFieldsFilter``) and of the
filters built on it or on the same idea:

  snow_depth_m                 R: filters/fields/snow_depth_m.py
  snow_cover                   R: filters/fields/snow_cover.py
  cos_sin_from_rad             R: filters/fields/cos_sin_from_rad.py
  cos_sin_mean_wave_direction  R: filters/fields/cos_sin_mean_wave_direction.py
  w_to_wz / wz_to_w            R: filters/fields/w_to_wz.py
  sum                          R: filters/fields/sum.py

``MatchingFieldsFilter`` keeps the reference behaviour for user-defined subclasses (one
``forward_transform(**fields)`` call per group, numpy inside).  The built-in filters derive
from ``StackMatchingFilter``: the grouping bookkeeping is the same, but the i-th operand of
every group is stacked into one HBM stack per operand and the expression is evaluated by a
single ``atx_combine_stack`` launch.  Output order equals the reference's: unmatched fields
first (in input order), then per group the returned inputs followed by the results.
"""

from __future__ import annotations

import logging
from abc import abstractmethod
from collections import defaultdict
from inspect import signature
from itertools import chain
from typing import Any, Callable, Iterable, Iterator

import numpy as np
import torch

from .. import native
from ..core import Filter, filter_registry, say_once
from ..fields import FieldList, fields_to_stack, new_field_from_numpy, new_field_from_stack, new_fieldlist_from_list
from ..grouping import GroupByParam, GroupByParamVertical

LOG = logging.getLogger(__name__)


def _names(value: Any) -> tuple[str, ...]:
    """A single name or any iterable of names, as a tuple."""
    if isinstance(value, str):
        return (value,)
    try:
        return tuple(value)
    except TypeError as e:
        raise TypeError(f"Expected str or iterable, got {type(value)}") from e


class MatchingSpec:
    """Which constructor arguments of a filter name its forward / backward operands, and which of the matched input
    fields are passed on next to the results (R: matching.py:35-81 — same arguments, same errors).

    Immutable value object: ``forward`` / ``backward`` are tuples of argument names; ``return_inputs`` is ``"all"``,
    ``"none"`` or a tuple of operand names (a subset of forward + backward); ``vertical`` groups by level type too.
    """

    __slots__ = ("select", "forward", "backward", "return_inputs", "vertical")
    _KEYWORDS = ("all", "none")

    def __init__(self, select: str = "param", forward: Any = (), backward: Any = (), return_inputs: Any = "none",
                 vertical: bool = False) -> None:
        if select != "param":
            raise NotImplementedError("Only 'select=param' is supported for now.")
        fwd, bwd = _names(forward), _names(backward)
        kept = return_inputs if return_inputs in self._KEYWORDS else _names(return_inputs)
        if kept not in self._KEYWORDS:
            operands = set(fwd) | set(bwd)
            if set(kept) - operands:
                raise ValueError(f"Returned input names must subset {operands}")
        for name, value in zip(self.__slots__, (select, fwd, bwd, kept, bool(vertical))):
            object.__setattr__(self, name, value)

    def __setattr__(self, name: str, value: Any) -> None:
        raise AttributeError(f"MatchingSpec is immutable (tried to set {name!r})")

    def _key(self) -> tuple:
        return tuple(getattr(self, name) for name in self.__slots__)

    def __eq__(self, other: Any) -> bool:
        return isinstance(other, MatchingSpec) and self._key() == other._key()

    def __hash__(self) -> int:
        return hash(self._key())

    def __repr__(self) -> str:
        return "MatchingSpec(" + ", ".join(f"{name}={getattr(self, name)!r}" for name in self.__slots__) + ")"

    def update_return_inputs(self, return_inputs: Any) -> "MatchingSpec":
        """This spec with another ``return_inputs`` (``self`` if nothing changes)."""
        kept = return_inputs if return_inputs in self._KEYWORDS else _names(return_inputs)
        if kept == self.return_inputs:
            return self
        return MatchingSpec(self.select, self.forward, self.backward, kept, self.vertical)

    def inputs(self, direction: str) -> tuple[str, ...]:
        """Names of the operands of ``direction`` ("forward" / "backward") whose fields are returned with the results."""
        if self.return_inputs == "none":
            return ()
        return tuple(getattr(self, direction)) if self.return_inputs == "all" else self.return_inputs


def inputs_generator(input_list: Iterable[str], **kwargs: Any) -> Iterator[Any]:
    for name in input_list:
        if name in kwargs:
            yield kwargs[name]


class MatchingFieldsFilter(Filter):
    """Converts groups of fields matched by their metadata (R: matching.py:90-311)."""

    MATCHING: MatchingSpec
    #: set by the filters whose arithmetic lives in a third-party package that is absent here (earthkit-meteo): what the device
    #: operator restates and what pins it — or where a statement of the reference makes the last bit of a library function visible
    #: (snow_cover).  Logged at WARNING by the first instance of the class in a process, at DEBUG by later ones (core.say_once).
    PARITY_NOTE: str | None = None

    @staticmethod
    def _check_expected_method_parameters(method: Callable, expected: set[str]) -> None:
        missing = set(expected) - set(signature(method).parameters)
        if missing:
            raise ValueError(f"{method}: missing parameters {missing}")

    def __init_subclass__(cls, **kwargs: Any) -> None:
        super().__init_subclass__(**kwargs)
        if cls.__dict__.get("_ABSTRACT_MATCHING_BASE"):
            return
        if not hasattr(cls, "MATCHING") or not isinstance(cls.MATCHING, MatchingSpec):
            raise TypeError(f"Class {cls.__name__} must define a 'MATCHING' attribute of type MatchingSpec.")
        fwd, bwd = set(cls.MATCHING.forward), set(cls.MATCHING.backward)
        MatchingFieldsFilter._check_expected_method_parameters(cls.__init__, fwd | bwd)
        MatchingFieldsFilter._check_expected_method_parameters(cls.forward_transform, fwd)
        MatchingFieldsFilter._check_expected_method_parameters(cls.backward_transform, bwd)

    def __init__(self, *args: Any, **kwargs: Any) -> None:
        super().__init__(*args, **kwargs)
        self._prepare_matching()
        if self.PARITY_NOTE:
            say_once(LOG, (type(self), "parity"), "%s: %s", type(self).__name__, self.PARITY_NOTE)

    def _prepare_matching(self) -> None:
        """Apply an instance-level ``return_inputs`` to the class's spec and warn about returned inputs that are not
        operands of a direction (they could never be returned there)."""
        wanted = getattr(self, "return_inputs", None)
        if wanted is not None:
            self.MATCHING = self.MATCHING.update_return_inputs(wanted)
        spec = self.MATCHING
        for direction, operands in (("forward", spec.forward), ("backward", spec.backward)):
            unknown = set(spec.inputs(direction)) - set(operands)
            if operands and unknown:
                LOG.warning(f"Some {direction} inputs will not be returned because they are not in the filter parameters: {unknown}")

    def _check_metadata_match(self, data: set, args: Iterable[str]) -> None:
        if not set(args).issubset(data):
            LOG.warning(
                "Please ensure your filter is configured to match the input variables metadata "
                f"current mismatch between inputs {data} and filter metadata {list(args)}"
            )

    def _grouping(self, group_by: tuple[str, ...]):
        return GroupByParamVertical(group_by) if self.MATCHING.vertical else GroupByParam(group_by)

    def forward(self, data: Any) -> FieldList:
        names = self.MATCHING.forward

        def transform(*fields: Any) -> Iterator[Any]:
            kwargs = dict(zip(names, fields, strict=True))
            return chain(inputs_generator(self.MATCHING.inputs(direction="forward"), **kwargs), self.forward_transform(**kwargs))

        return self._transform(data, transform, *(getattr(self, n) for n in names))

    def backward(self, data: Any) -> FieldList:
        names = self.MATCHING.backward

        def transform(*fields: Any) -> Iterator[Any]:
            kwargs = dict(zip(names, fields, strict=True))
            return chain(inputs_generator(self.MATCHING.inputs(direction="backward"), **kwargs), self.backward_transform(**kwargs))

        return self._transform(data, transform, *(getattr(self, n) for n in names))

    def _transform(self, data: Any, transform: Callable[..., Iterator[Any]], *group_by: str) -> FieldList:
        data = data if isinstance(data, FieldList) else FieldList(list(data))
        self._check_metadata_match(set(data.metadata(self.MATCHING.select)), group_by)
        result: list[Any] = []
        for matching in self._grouping(group_by).iterate(data, other=result.append):
            for f in transform(*matching):
                result.append(f)
        return self.new_fieldlist_from_list(result)

    def new_field_from_numpy(self, array: np.ndarray, *, template: Any, **kwargs: Any) -> Any:
        return new_field_from_numpy(array, template=template, **kwargs)

    def new_fieldlist_from_list(self, fields: list[Any]) -> FieldList:
        return new_fieldlist_from_list(fields)

    @abstractmethod
    def forward_transform(self, *fields: Any) -> Iterator[Any]: ...

    def backward_transform(self, *fields: Any) -> Iterator[Any]:
        raise NotImplementedError("Backward transformation not implemented.")


MatchingFieldsFilter._ABSTRACT_MATCHING_BASE = True


# =================================================================================
# stack-level execution
# =================================================================================
def combine_groups(
    groups: list[tuple[Any, ...]],
    op: int,
    n_out: int,
    *,
    flags: int = 0,
    level_values: list[float] | None = None,
    check: Callable[[list[Any]], None] | None = None,
    after: Callable[[list[Any]], None] | None = None,
) -> list[list[tuple[Any, int]]]:
    """Evaluate one operator for all groups: returns, per group, the ``(stack, level)`` of each output.

    Groups are bucketed by grid size; operand ``i`` of every group of a bucket becomes level
    ``g`` of the i-th input stack; one ``atx_combine_stack`` launch per bucket.
    """
    results: list[list[tuple[Any, int]]] = [[] for _ in groups]
    buckets: dict[int, list[int]] = defaultdict(list)
    for gi, g in enumerate(groups):
        buckets[int(np.prod(g[0].shape))].append(gi)
    for members in buckets.values():
        n_in = len(groups[members[0]])
        ins = [fields_to_stack([groups[gi][i] for gi in members]) for i in range(n_in)]
        dtype = torch.float32 if all(s.dtype == torch.float32 for s in ins) else torch.float64
        ins = [s if s.dtype == dtype else type(s)(s.data.to(dtype), s.n_pts, s.n_lev, s.layout) for s in ins]
        if check is not None:
            check(ins)
        outs = [ins[0].new_like(zero=False) for _ in range(n_out)]
        level_param = None
        if level_values is not None:
            level_param = torch.tensor([level_values[gi] for gi in members], dtype=torch.float64, device=ins[0].device)
        native.combine_stack(op, [s.data for s in ins], [s.data for s in outs], n_pts=ins[0].n_pts, n_lev=ins[0].n_lev,
                             pitch=ins[0].pitch, layout=ins[0].layout, level_param=level_param, flags=flags)
        if after is not None:
            after(outs)
        for level, gi in enumerate(members):
            results[gi] = [(o, level) for o in outs]
    return results


class StackMatchingFilter(MatchingFieldsFilter):
    """A ``MatchingFieldsFilter`` whose per-group expression is one ``atx_combine_stack`` operator.

    Subclasses describe each direction with ``<direction>_plan(**fields)`` returning
    ``(op, flags, [(template_field, metadata), ...], level_value)`` for ONE group; the
    operator and flags must not depend on the group.  A fifth element, if present, lists input
    fields handed on unchanged AFTER the results of the group.
    """

    _ABSTRACT_MATCHING_BASE = True

    def forward_plan(self, **fields: Any):
        raise NotImplementedError

    def backward_plan(self, **fields: Any):
        raise NotImplementedError("Backward transformation not implemented.")

    def _check_stacks(self, direction: str, stacks: list[Any]) -> None:
        pass

    def _check_results(self, direction: str, stacks: list[Any]) -> None:
        pass

    def _run(self, data: Any, direction: str) -> FieldList:
        names = getattr(self.MATCHING, direction)
        plan_fn = self.forward_plan if direction == "forward" else self.backward_plan
        group_by = tuple(getattr(self, n) for n in names)
        data = data if isinstance(data, FieldList) else FieldList(list(data))
        self._check_metadata_match(set(data.metadata(self.MATCHING.select)), group_by)
        result: list[Any] = []
        groups = list(self._grouping(group_by).iterate(data, other=result.append))
        if not groups:
            return self.new_fieldlist_from_list(result)
        plans = [plan_fn(**dict(zip(names, g, strict=True))) for g in groups]
        op, flags = plans[0][0], plans[0][1]
        level_values = [p[3] for p in plans] if plans[0][3] is not None else None
        outs = combine_groups(groups, op, len(plans[0][2]), flags=flags, level_values=level_values,
                              check=lambda stacks: self._check_stacks(direction, stacks),
                              after=lambda stacks: self._check_results(direction, stacks))
        for g, plan, out in zip(groups, plans, outs):
            kwargs = dict(zip(names, g, strict=True))
            result.extend(inputs_generator(self.MATCHING.inputs(direction=direction), **kwargs))
            for (template, metadata), (stack, level) in zip(plan[2], out):
                result.append(new_field_from_stack(stack, level, template=template, metadata=metadata))
            result.extend(plan[4] if len(plan) > 4 else ())
        return self.new_fieldlist_from_list(result)

    def forward(self, data: Any) -> FieldList:
        return self._run(data, "forward")

    def backward(self, data: Any) -> FieldList:
        return self._run(data, "backward")

    # the per-group entry points of the reference API, for callers that use them directly
    def forward_transform(self, *args: Any, **fields: Any) -> Iterator[Any]:
        fields.update(dict(zip(self.MATCHING.forward, args)))
        op, flags, outputs, level, *rest = self.forward_plan(**fields)
        group = tuple(fields[n] for n in self.MATCHING.forward)
        (out,) = combine_groups([group], op, len(outputs), flags=flags, level_values=None if level is None else [level],
                                check=lambda stacks: self._check_stacks("forward", stacks),
                                after=lambda stacks: self._check_results("forward", stacks))
        for (template, metadata), (stack, lvl) in zip(outputs, out):
            yield new_field_from_stack(stack, lvl, template=template, metadata=metadata)
        yield from (rest[0] if rest else ())

    def backward_transform(self, *args: Any, **fields: Any) -> Iterator[Any]:
        fields.update(dict(zip(self.MATCHING.backward, args)))
        op, flags, outputs, level, *_ = self.backward_plan(**fields)
        group = tuple(fields[n] for n in self.MATCHING.backward)
        (out,) = combine_groups([group], op, len(outputs), flags=flags, level_values=None if level is None else [level])
        for (template, metadata), (stack, lvl) in zip(outputs, out):
            yield new_field_from_stack(stack, lvl, template=template, metadata=metadata)


# =================================================================================
# the filters
# =================================================================================
@filter_registry.register("snow_depth_m")
class SnowDepthM(StackMatchingFilter):
    """Snow depth in metres: ``sde = 1000 * sd / rsn``."""

    MATCHING = MatchingSpec(select="param", forward=("snow_depth", "snow_density"))

    def __init__(self, *, snow_depth: str = "sd", snow_density: str = "rsn", snow_depth_m: str = "sde") -> None:
        self.snow_depth = snow_depth
        self.snow_density = snow_density
        self.snow_depth_m = snow_depth_m
        super().__init__()

    def forward_plan(self, snow_depth: Any, snow_density: Any):
        return native.COMB_SNOW_DEPTH_M, 0, [(snow_depth, dict(param=self.snow_depth_m, units="m"))], None

    def forward_transform(self, snow_depth: Any = None, snow_density: Any = None) -> Iterator[Any]:
        return super().forward_transform(snow_depth=snow_depth, snow_density=snow_density)


@filter_registry.register("snow_cover")
class SnowCover(StackMatchingFilter):
    """Snow cover fraction from snow depth and snow density (R: snow_cover.py:34-39)."""

    MATCHING = MatchingSpec(select="param", forward=("snow_depth", "snow_density"))
    PARITY_NOTE = ("values agree with the reference's numpy statement to 1e-6 relative EXCEPT at the statement's own jump, `snow_cover[snow_cover > "
                   "0.99] = 1.0` (snow_cover.py:38): where tanh lands within a few ulp of 0.99 the last bit of tanh decides between 0.99 and 1.0, "
                   "and the device's tanh and numpy's may differ in that bit — either side is the statement's own value (about one point in "
                   "30 000 in float32, none in practice in float64)")

    def __init__(self, *, snow_depth: str = "sd", snow_density: str = "rsn", snow_cover: str = "snowc") -> None:
        self.snow_depth = snow_depth
        self.snow_density = snow_density
        self.snow_cover = snow_cover
        super().__init__()

    def forward_plan(self, snow_depth: Any, snow_density: Any):
        return native.COMB_SNOW_COVER, 0, [(snow_depth, dict(param=self.snow_cover, units="Fraction"))], None

    def forward_transform(self, snow_depth: Any = None, snow_density: Any = None) -> Iterator[Any]:
        return super().forward_transform(snow_depth=snow_depth, snow_density=snow_density)


def _range_check(name: str, stack: Any) -> None:
    """R: cos_sin_from_rad.py:73-76 — radians expected in [-2 pi, 2 pi]; min / max by ``atx_reduce``."""
    kw = dict(n_pts=stack.n_pts, n_lev=stack.n_lev, pitch=stack.pitch, layout=stack.layout)
    lo, hi = native.reduce_stack(stack.data, native.RED_MINMAX, **kw)  # one pass, one read-back
    if lo < -2 * np.pi:
        raise ValueError(f"Param {name} is expected in radians in the range [-2pi, pi], but min={lo}")
    if hi > 2 * np.pi:
        raise ValueError(f"Param {name} is expected in radians in the range [-2pi, pi], but max={hi}")


@filter_registry.register("cos_sin_from_rad")
class CosSinFromRad(StackMatchingFilter):
    """A direction in radians -> its cosine and sine, and back with atan2."""

    MATCHING = MatchingSpec(select="param", forward=("param",), backward=("cos_param", "sin_param"))

    def __init__(self, *, param: str, cos_param: str | None = None, sin_param: str | None = None) -> None:
        self.param = param
        self.cos_param = cos_param if cos_param is not None else f"cos_{param}"
        self.sin_param = sin_param if sin_param is not None else f"sin_{param}"
        super().__init__()

    def forward_plan(self, param: Any):
        return native.COMB_COS_SIN, 0, [(param, dict(param=self.cos_param)), (param, dict(param=self.sin_param))], None

    def backward_plan(self, cos_param: Any, sin_param: Any):
        return native.COMB_ATAN2, 0, [(cos_param, dict(param=self.param))], None

    def _check_stacks(self, direction: str, stacks: list[Any]) -> None:
        if direction == "forward":
            _range_check(self.param, stacks[0])

    def forward_transform(self, param: Any = None) -> Iterator[Any]:
        return super().forward_transform(param=param)

    def backward_transform(self, cos_param: Any = None, sin_param: Any = None) -> Iterator[Any]:
        return super().backward_transform(cos_param=cos_param, sin_param=sin_param)

    def patch_data_request(self, data_request: dict[str, Any]) -> dict[str, Any]:
        param = data_request.get("param")
        if param is None:
            return data_request
        if self.cos_param in param or self.sin_param in param:
            data_request["param"] = [p for p in param if p not in (self.cos_param, self.sin_param)]
            data_request["param"].append(self.param)
        return data_request


@filter_registry.register("cos_sin_mean_wave_direction")
class CosSinWaveDirection(StackMatchingFilter):
    """Mean wave direction in degrees -> cosine / sine, and back to [0, 360)."""

    MATCHING = MatchingSpec(
        select="param", forward=("mean_wave_direction",), backward=("cos_mean_wave_direction", "sin_mean_wave_direction")
    )

    def __init__(self, *, mean_wave_direction: str = "mwd", cos_mean_wave_direction: str = "cos_mwd",
                 sin_mean_wave_direction: str = "sin_mwd") -> None:
        self.mean_wave_direction = mean_wave_direction
        self.cos_mean_wave_direction = cos_mean_wave_direction
        self.sin_mean_wave_direction = sin_mean_wave_direction
        super().__init__()

    def forward_plan(self, mean_wave_direction: Any):
        return native.COMB_COS_SIN, native.COMB_DEGREES, [
            (mean_wave_direction, dict(param=self.cos_mean_wave_direction)),
            (mean_wave_direction, dict(param=self.sin_mean_wave_direction)),
        ], None

    def backward_plan(self, cos_mean_wave_direction: Any, sin_mean_wave_direction: Any):
        return native.COMB_ATAN2, native.COMB_DEGREES, [(cos_mean_wave_direction, dict(param=self.mean_wave_direction))], None

    def forward_transform(self, mean_wave_direction: Any = None) -> Iterator[Any]:
        return super().forward_transform(mean_wave_direction=mean_wave_direction)

    def backward_transform(self, cos_mean_wave_direction: Any = None, sin_mean_wave_direction: Any = None) -> Iterator[Any]:
        return super().backward_transform(cos_mean_wave_direction=cos_mean_wave_direction,
                                          sin_mean_wave_direction=sin_mean_wave_direction)

    def patch_data_request(self, data_request: dict[str, Any]) -> dict[str, Any]:
        param = data_request.get("param")
        if param is None:
            return data_request
        if self.cos_mean_wave_direction in param or self.sin_mean_wave_direction in param:
            data_request["param"] = [p for p in param if p not in (self.cos_mean_wave_direction, self.sin_mean_wave_direction)]
            data_request["param"].append(self.mean_wave_direction)
        return data_request


class VerticalVelocity(StackMatchingFilter):
    """Pressure vertical velocity w [Pa/s] <-> geometric vertical velocity wz [m/s] (hydrostatic, ideal gas)."""

    MATCHING = MatchingSpec(
        select="param",
        forward=("vertical_velocity", "temperature", "humidity"),
        backward=("geometric_vertical_velocity", "temperature", "humidity"),
    )

    def __init__(self, *, vertical_velocity: str = "w", geometric_vertical_velocity: str = "wz", temperature: str = "t",
                 humidity: str = "q", return_inputs: Any = "all") -> None:
        self.return_inputs = return_inputs
        self.vertical_velocity = vertical_velocity
        self.geometric_vertical_velocity = geometric_vertical_velocity
        self.temperature = temperature
        self.humidity = humidity
        super().__init__()

    @staticmethod
    def _level(field: Any) -> float:
        return float(field.metadata("levelist", default=None))  # R: w_to_wz.py:97 — TypeError if the level is missing

    def forward_plan(self, vertical_velocity: Any, temperature: Any, humidity: Any):
        return (native.COMB_W_TO_WZ, 0, [(vertical_velocity, dict(param=self.geometric_vertical_velocity))],
                self._level(vertical_velocity))

    def backward_plan(self, geometric_vertical_velocity: Any, temperature: Any, humidity: Any):
        return (native.COMB_WZ_TO_W, 0, [(geometric_vertical_velocity, dict(param=self.vertical_velocity))],
                self._level(geometric_vertical_velocity))

    def forward_transform(self, vertical_velocity: Any = None, temperature: Any = None, humidity: Any = None) -> Iterator[Any]:
        return super().forward_transform(vertical_velocity=vertical_velocity, temperature=temperature, humidity=humidity)

    def backward_transform(self, geometric_vertical_velocity: Any = None, temperature: Any = None, humidity: Any = None) -> Iterator[Any]:
        return super().backward_transform(geometric_vertical_velocity=geometric_vertical_velocity, temperature=temperature,
                                          humidity=humidity)


filter_registry.register("w_to_wz", VerticalVelocity)
filter_registry.register("wz_to_w", VerticalVelocity.reversed)


class WindComponents(StackMatchingFilter):
    """U and V wind components -> wind speed and direction (the direction the wind blows FROM, degrees clockwise from north), and back
    (R: filters/fields/uv_to_ddff.py:23-128).  The reference delegates the arithmetic to earthkit-meteo's ``xy_to_polar`` /
    ``polar_to_xy`` (absent here): its published "meteo" convention is restated in ``atx_combine_stack`` and pinned by the
    reference's own test vectors (tests/field_filters/test_uv_to_ddff.py)."""

    MATCHING = MatchingSpec(select="param", forward=("u_component", "v_component"), backward=("wind_speed", "wind_direction"))
    PARITY_NOTE = ("arithmetic restated from earthkit-meteo's wind.xy_to_polar / polar_to_xy (package absent here); pinned only at the "
                   "reference's test points (tests/field_filters/test_uv_to_ddff.py) at np.allclose, not bit for bit")

    def __init__(self, *, u_component: str = "u", v_component: str = "v", wind_speed: str = "ws", wind_direction: str = "wdir",
                 convention: str = "meteo", radians: bool = False) -> None:
        self.u_component, self.v_component = u_component, v_component
        self.wind_speed, self.wind_direction = wind_speed, wind_direction
        self.convention, self.radians = convention, radians
        assert not self.radians, "Radians not (yet) supported"  # R: uv_to_ddff.py:74
        if convention != "meteo":
            raise NotImplementedError(f"wind convention {convention!r}: only 'meteo' is implemented on the device")
        super().__init__()

    def forward_plan(self, u_component: Any, v_component: Any):
        return native.COMB_XY_TO_POLAR, 0, [(u_component, dict(param=self.wind_speed)), (v_component, dict(param=self.wind_direction))], None

    def backward_plan(self, wind_speed: Any, wind_direction: Any):
        return native.COMB_POLAR_TO_XY, 0, [(wind_speed, dict(param=self.u_component)), (wind_direction, dict(param=self.v_component))], None

    def forward_transform(self, u_component: Any = None, v_component: Any = None) -> Iterator[Any]:
        return super().forward_transform(u_component=u_component, v_component=v_component)

    def backward_transform(self, wind_speed: Any = None, wind_direction: Any = None) -> Iterator[Any]:
        return super().backward_transform(wind_speed=wind_speed, wind_direction=wind_direction)


filter_registry.register("uv_to_ddff", WindComponents)
filter_registry.register("ddff_to_uv", WindComponents.reversed)


@filter_registry.register("sum")
class Sum(Filter):
    """Replace ``params`` by their sum ``output``, per date / level / member (R: sum.py:25-121).

    Fields are matched on their MARS namespace minus ``param`` (minus ``levelist`` with
    ``ignore_level``); the terms are added in the order the fields appear in the input, as the
    reference does.
    """

    def __init__(self, *, params: list[str], output: str, ignore_level: bool = False) -> None:
        self.params = params
        self.output = output
        self.ignore_level = ignore_level

    def forward(self, fields: Any) -> FieldList:
        result: list[Any] = []
        needed: dict[tuple, dict[str, Any]] = defaultdict(dict)
        for f in fields:
            key = dict(f.metadata(namespace="mars"))
            param = key.pop("param", None)
            if self.ignore_level:
                key.pop("levelist", None)
            if param is None:
                param = f.metadata("param")
            if param in self.params:
                frozen = tuple(key.items())
                if param in needed[frozen]:
                    raise ValueError(f"Duplicate field {param} for {frozen}")
                needed[frozen][param] = f
            else:
                result.append(f)
        groups = []
        for values in needed.values():
            if len(values) != len(self.params):
                raise ValueError("Missing fields")
            groups.append(tuple(values.values()))
        if groups:
            if len(self.params) > native.COMB_MAX_INPUTS:
                raise NotImplementedError(f"sum of more than {native.COMB_MAX_INPUTS} fields is not supported")
            outs = combine_groups(groups, native.COMB_SUM, 1)
            for g, out in zip(groups, outs):
                stack, level = out[0]
                # R: sum.py:109-117 — the sum is a flattened array; the first term is the template
                field = new_field_from_stack(stack, level, template=g[0], metadata=dict(param=self.output))
                field.shape = (stack.n_pts,)
                result.append(field)
        return new_fieldlist_from_list(result)

    def backward(self, data: Any) -> Any:
        raise NotImplementedError("Sum filter is not reversible")


@filter_registry.register("accum_to_interval")
class AccumToInterval(Filter):
    """Accumulated-since-start fields -> per-interval values by differencing consecutive valid times
    (R: filters/fields/accum_to_interval.py:24-101).  Fields are grouped by (param, level, levelType) and
    sorted by ``valid_datetime``; the first of each selected group becomes zeros (``zero_left``) or is kept.
    All differences of all groups are ONE ``atx_combine_stack`` launch (current stack minus previous stack)."""

    def __init__(self, variables: Iterable[str], window: str | None = None, zero_left: bool = True, **kwargs: Any) -> None:
        self.variables = set(variables)
        self.zero_left = bool(zero_left)
        self.window = window  # accepted for YAML recipes, not used by the algorithm

    @staticmethod
    def _identifier(f: Any) -> tuple:
        return (f.metadata("param"), f.metadata("level", default=None), f.metadata("levelType", default=None))

    def forward(self, fields: Any) -> FieldList:
        groups: dict[tuple, list[Any]] = {}
        for f in fields:
            groups.setdefault(self._identifier(f), []).append(f)
        for key, members in groups.items():
            groups[key] = sorted(members, key=lambda x: x.metadata("valid_datetime"))

        pairs = []  # (current, previous) of every selected group, in output order
        for (param, _, _), members in groups.items():
            if param in self.variables:
                pairs.extend((members[i], members[i - 1]) for i in range(1, len(members)))
        diffs = iter(combine_groups(pairs, native.COMB_SUB, 1)) if pairs else iter(())

        out: list[Any] = []
        for (param, _, _), members in groups.items():
            if param not in self.variables or len(members) == 0:
                out.extend(members)
                continue
            first = members[0]
            if self.zero_left:
                out.append(new_field_from_numpy(np.zeros(first.shape), template=first))  # np.zeros_like(first.to_numpy())
            else:
                out.append(first)
            for i in range(1, len(members)):
                stack, level = next(diffs)[0]
                out.append(new_field_from_stack(stack, level, template=members[i]))
        return new_fieldlist_from_list(out)
