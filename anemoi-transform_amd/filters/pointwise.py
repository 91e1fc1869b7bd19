"""Per-point variable / unit transforms as stack-level kernel launches.

Each class keeps the reference filter's registered name, inputs, selection and
metadata changes; only the arithmetic moves from a per-field numpy statement to
one ``atx_pointwise_stack`` launch per stack (``engine.run_level_ops``):

  rescale / convert     R: filters/fields/rescale.py:19-111
  orog_to_z / z_to_orog R: filters/fields/orog_to_z.py:19-98
  clip_fields           R: filters/fields/clipper.py:18-70
  impute_nans_fields    R: filters/fields/impute_nans.py:21-55
  lnsp_to_sp / sp_to_lnsp R: filters/fields/lnsp_to_sp.py:19-97
  glacier_mask          R: filters/fields/glacier_mask.py:39-67
  noop                  R: filters/fields/noop.py
"""

from __future__ import annotations

import logging
from typing import Any

import numpy as np
import torch

from .. import native
from ..core import Filter, SingleFieldFilter, filter_registry, say_once
from ..fields import FieldList
from .engine import LevelOp, PointMask, run_level_ops

LOG = logging.getLogger(__name__)

# R: constants.py:13 (earthkit.meteo g); value pinned by R: filters/tabular/geopotential_to_height.py:51
g_gravitational_acceleration = 9.80665


class StackFieldFilter(SingleFieldFilter):
    """A ``SingleFieldFilter`` whose transform is one per-level operator of libatx."""

    def forward_level_op(self, field: Any) -> LevelOp:
        raise NotImplementedError

    def backward_level_op(self, field: Any) -> LevelOp:
        raise NotImplementedError("Field backward transform not implemented.")  # R: filter.py:160

    def forward_metadata(self, field: Any) -> dict[str, Any]:
        return {}

    def backward_metadata(self, field: Any) -> dict[str, Any]:
        return {}

    def point_mask(self) -> PointMask | None:
        return None

    def forward(self, data: Any) -> FieldList:
        return run_level_ops(data, self._forward_selection.match, self.forward_level_op, self.forward_metadata, self.point_mask())

    def backward(self, data: Any) -> FieldList:
        return run_level_ops(data, self._backward_selection.match, self.backward_level_op, self.backward_metadata, self.point_mask())

    # single-field entry points of the reference API (selection already decided by the caller)
    def forward_transform(self, field: Any) -> Any:
        return run_level_ops([field], lambda f: True, self.forward_level_op, self.forward_metadata, self.point_mask())[0]

    def backward_transform(self, field: Any) -> Any:
        return run_level_ops([field], lambda f: True, self.backward_level_op, self.backward_metadata, self.point_mask())[0]


# ---------------------------------------------------------------------------------
# rescale / convert
# ---------------------------------------------------------------------------------
class Rescaler:
    """``x*scale + offset`` and its inverse (R: rescale.py:19-28) as level operators."""

    def __init__(self, scale: float, offset: float) -> None:
        self.scale = scale
        self.offset = offset

    def forward_op(self) -> LevelOp:
        return (native.OP_AFFINE, 0, float(self.scale), float(self.offset))

    def backward_op(self) -> LevelOp:
        return (native.OP_AFFINE_INV, 0, float(self.scale), float(self.offset))


class RescaleMixin:
    forward_units = None
    backward_units = None

    def forward_select(self):
        return {"param": self.param}

    def forward_level_op(self, field: Any) -> LevelOp:
        return self.rescaler.forward_op()

    def backward_level_op(self, field: Any) -> LevelOp:
        return self.rescaler.backward_op()

    def forward_metadata(self, field: Any) -> dict[str, Any]:
        return dict(param=self.param, units=self.forward_units)  # R: rescale.py:52

    def backward_metadata(self, field: Any) -> dict[str, Any]:
        return dict(param=self.param)  # R: rescale.py:57


class Rescale(RescaleMixin, StackFieldFilter):
    """Rescale a parameter by a scale and an offset, and back."""

    required_inputs = ("scale", "offset", "param")

    def prepare_filter(self):
        self.rescaler = Rescaler(self.scale, self.offset)


# unit -> (factor, offset) with  base = x*factor + offset.  Used only when pint is not
# installed; K -> degC is the one pair the reference pins (R: tests/field_filters/test_rescale.py:58-72).
_UNITS: dict[str, tuple[str, float, float]] = {
    "K": ("temperature", 1.0, 0.0), "kelvin": ("temperature", 1.0, 0.0),
    "degC": ("temperature", 1.0, 273.15), "celsius": ("temperature", 1.0, 273.15), "degree_Celsius": ("temperature", 1.0, 273.15),
    "degF": ("temperature", 5.0 / 9.0, 459.67 * 5.0 / 9.0), "fahrenheit": ("temperature", 5.0 / 9.0, 459.67 * 5.0 / 9.0),
    "Pa": ("pressure", 1.0, 0.0), "pascal": ("pressure", 1.0, 0.0), "hPa": ("pressure", 100.0, 0.0),
    "mbar": ("pressure", 100.0, 0.0), "kPa": ("pressure", 1000.0, 0.0), "bar": ("pressure", 1e5, 0.0),
    "m": ("length", 1.0, 0.0), "meter": ("length", 1.0, 0.0), "km": ("length", 1000.0, 0.0),
    "cm": ("length", 0.01, 0.0), "mm": ("length", 0.001, 0.0),
    "kg": ("mass", 1.0, 0.0), "g": ("mass", 0.001, 0.0),
    "s": ("time", 1.0, 0.0), "min": ("time", 60.0, 0.0), "h": ("time", 3600.0, 0.0), "hour": ("time", 3600.0, 0.0),
    "m/s": ("speed", 1.0, 0.0), "km/h": ("speed", 1000.0 / 3600.0, 0.0), "knot": ("speed", 1852.0 / 3600.0, 0.0),
    "1": ("fraction", 1.0, 0.0), "fraction": ("fraction", 1.0, 0.0), "%": ("fraction", 0.01, 0.0), "percent": ("fraction", 0.01, 0.0),
}


def _have_pint() -> bool:
    try:
        import pint  # noqa: F401
    except ImportError:
        return False
    return True


def _convert_value(x: float, unit_in: str, unit_out: str) -> float:
    try:
        import pint  # the reference's converter (R: rescale.py:94), if installed

        return pint.UnitRegistry().Quantity(x, unit_in).to(unit_out).magnitude
    except ImportError:
        pass
    if unit_in not in _UNITS or unit_out not in _UNITS:
        raise ValueError(f"convert: unknown unit {unit_in!r} or {unit_out!r} (pint is not installed; known: {sorted(_UNITS)})")
    (dim_i, f_i, o_i), (dim_o, f_o, o_o) = _UNITS[unit_in], _UNITS[unit_out]
    if dim_i != dim_o:
        raise ValueError(f"convert: cannot convert {unit_in!r} ({dim_i}) to {unit_out!r} ({dim_o})")
    if f_i == f_o:
        return x + (o_i - o_o) / f_o if o_i != o_o else x
    return (x * f_i + o_i - o_o) / f_o


class Convert(RescaleMixin, StackFieldFilter):
    """Convert a parameter from one unit to another, and back (scale / offset derived as in R: rescale.py:93-107)."""

    required_inputs = ("unit_in", "unit_out", "param")

    def prepare_filter(self):
        self.forward_units = self.unit_out
        self.backward_units = self.unit_in
        x1, x2 = 0.0, 1.0
        y1 = _convert_value(x1, self.unit_in, self.unit_out)
        y2 = _convert_value(x2, self.unit_in, self.unit_out)
        scale = (y2 - y1) / (x2 - x1)
        offset = y1 - scale * x1
        self.rescaler = Rescaler(scale, offset)
        if {self.unit_in, self.unit_out} != {"K", "degC"}:
            # the one pair the reference pins is K -> degC (R: tests/field_filters/test_rescale.py:58-72); everything else comes from
            # pint when it is installed, else from the private table above — say so, as `regrid`'s default route does
            # (pint IS the reference's own converter, R: rescale.py:94: with it installed this is information, not a warning; either way
            # once per unit pair and process — core.say_once — not once per construction)
            say_once(LOG, ("convert", self.unit_in, self.unit_out),
                     "convert(%s -> %s): scale %r and offset %r come from %s; only K <-> degC is pinned by the reference "
                     "(tests/field_filters/test_rescale.py), other pairs are not held to a reference vector",
                     self.unit_in, self.unit_out, scale, offset,
                     "pint, the reference's own converter" if _have_pint() else "this package's private unit table (pint is not installed)",
                     level=logging.INFO if _have_pint() else logging.WARNING)


filter_registry.register("rescale", Rescale)
filter_registry.register("convert", Convert)


# ---------------------------------------------------------------------------------
# orography <-> geopotential
# ---------------------------------------------------------------------------------
class Orography(StackFieldFilter):
    """Orography [m] x g <-> surface geopotential [m2/s2] / g."""

    optional_inputs = {"orography": "orog", "geopotential": "z"}

    def forward_select(self):
        return {"param": self.orography}

    def backward_select(self):
        return {"param": self.geopotential}

    def forward_level_op(self, field: Any) -> LevelOp:
        return (native.OP_MUL, 0, g_gravitational_acceleration, 0.0)  # R: orog_to_z.py:59

    def backward_level_op(self, field: Any) -> LevelOp:
        return (native.OP_DIV, 0, g_gravitational_acceleration, 0.0)  # R: orog_to_z.py:77 — a division

    def forward_metadata(self, field: Any) -> dict[str, Any]:
        return {"param": self.geopotential}

    def backward_metadata(self, field: Any) -> dict[str, Any]:
        return {"param": self.orography}

    def patch_data_request(self, data_request: Any) -> Any:
        # R: orog_to_z.py:80-94
        param = data_request.get("param")
        if param is None:
            return data_request
        param = param if isinstance(param, list) else [param]
        if self.geopotential in param and self.orography in param:
            raise ValueError("Data request cannot contain both orography and geopotential parameters.")
        on_levels = data_request.get("levtype", "") == "pl" or data_request.get("levelist", [])
        if self.geopotential in param and on_levels:
            data_request["param"] = [self.orography if p == self.geopotential else p for p in param]
        elif self.orography in param and on_levels:
            data_request["param"] = [self.geopotential if p == self.orography else p for p in param]
        return data_request


filter_registry.register("orog_to_z_fields", Orography)
filter_registry.register("z_to_orog_fields", Orography.reversed)


# ---------------------------------------------------------------------------------
# clip / impute / lnsp
# ---------------------------------------------------------------------------------
@filter_registry.register("clip_fields")
class Clipper(StackFieldFilter):
    """Clip one parameter to ``[minimum, maximum]`` (either bound optional)."""

    required_inputs = ("param",)
    optional_inputs = {"minimum": None, "maximum": None}

    def prepare_filter(self):
        if self.minimum is None and self.maximum is None:
            raise ValueError("At least one value for minimum or maximum must be specified.")

    def forward_select(self):
        return {"param": self.param}

    def forward_level_op(self, field: Any) -> LevelOp:
        nan = float("nan")  # a NaN bound = "no bound on that side" (np.clip(x, None, hi))
        return (native.OP_CLIP, 0, nan if self.minimum is None else float(self.minimum),
                nan if self.maximum is None else float(self.maximum))

    def forward_metadata(self, field: Any) -> dict[str, Any]:
        return {"param": self.param}


@filter_registry.register("impute_nans_fields")
class ImputeNaNs(StackFieldFilter):
    """Replace NaNs of the listed parameters by a fixed value."""

    required_inputs = ("param", "value")

    def forward_select(self):
        return {"param": self.param}

    def forward_level_op(self, field: Any) -> LevelOp:
        return (native.OP_IMPUTE_NAN, 0, float(self.value), 0.0)


class LnspToSp(StackFieldFilter):
    """ln(surface pressure) <-> surface pressure."""

    optional_inputs = {"log_of_surface_pressure": "lnsp", "surface_pressure": "sp"}

    def forward_select(self):
        return {"param": self.log_of_surface_pressure}

    def backward_select(self):
        return {"param": self.surface_pressure}

    def forward_level_op(self, field: Any) -> LevelOp:
        return (native.OP_EXP, 0, 0.0, 0.0)

    def backward_level_op(self, field: Any) -> LevelOp:
        return (native.OP_LOG, 0, 0.0, 0.0)

    def forward_metadata(self, field: Any) -> dict[str, Any]:
        return {"param": self.surface_pressure, "levelist": None, "level": None}  # R: lnsp_to_sp.py:45

    def backward_metadata(self, field: Any) -> dict[str, Any]:
        return {"param": self.log_of_surface_pressure}

    def patch_data_request(self, data_request: dict[str, Any]) -> dict[str, Any]:
        # R: lnsp_to_sp.py:68-97: swap the requested parameter for its counterpart
        param = data_request.get("param")
        if param is None:
            return data_request
        param = param if isinstance(param, list) else [param]
        if self.surface_pressure in param and self.log_of_surface_pressure in param:
            raise ValueError("Data request cannot contain both surface pressure and log of surface pressure parameters.")
        if self.surface_pressure in param:
            data_request["param"].remove(self.surface_pressure)
            data_request["param"].append(self.log_of_surface_pressure)
        elif self.log_of_surface_pressure in param:
            data_request["param"].remove(self.log_of_surface_pressure)
            data_request["param"].append(self.surface_pressure)
        return data_request


filter_registry.register("lnsp_to_sp", LnspToSp)
filter_registry.register("sp_to_lnsp", LnspToSp.reversed)


# ---------------------------------------------------------------------------------
# glacier mask: NaN where a (file) mask is set
# ---------------------------------------------------------------------------------
def load_mask_file(path: str) -> np.ndarray:
    """First field of a mask file, flattened.  ``.npy`` / ``.npz`` natively; anything else needs earthkit-data
    (R: apply_mask.py:153-158, glacier_mask.py:46)."""
    if path.endswith(".npy"):
        return np.load(path)
    if path.endswith(".npz"):
        loaded = np.load(path)
        return loaded["mask"] if "mask" in loaded else loaded[list(loaded.keys())[0]]
    try:
        import earthkit.data as ekd  # type: ignore
    except ImportError as e:
        raise ImportError(f"reading {path!r} needs earthkit-data, which is not installed; use a .npy / .npz mask") from e
    return ekd.from_source("file", path)[0].to_numpy(flatten=True)


@filter_registry.register("glacier_mask")
class SnowDepthMasked(StackFieldFilter):
    """Mask out glaciers in snow depth (NaN where the glacier mask is true)."""

    required_inputs = ("glacier_mask",)
    optional_inputs = {"snow_depth": "sd", "snow_depth_masked": "sd_masked"}

    def prepare_filter(self):
        self._mask_host = np.asarray(load_mask_file(self._config["glacier_mask"])).astype(bool).reshape(-1)
        self._mask_dev: PointMask | None = None

    def point_mask(self) -> PointMask:
        if self._mask_dev is None:
            from ..stack import device

            t = torch.from_numpy(np.pad(self._mask_host.astype(np.uint8), (0, 8))).to(device())
            self._mask_dev = PointMask(t, self._mask_host.size)
        return self._mask_dev

    def forward_select(self):
        return {"param": self.snow_depth}

    def forward_level_op(self, field: Any) -> LevelOp:
        return (native.OP_COPY, 1, 0.0, 0.0)  # R: glacier_mask.py:33 snow_depth[glacier_mask] = nan

    def forward_metadata(self, field: Any) -> dict[str, Any]:
        return dict(param=self.snow_depth_masked, units="Fraction")


@filter_registry.register("noop")
class NoOp(Filter):
    """Returns its input unchanged."""

    def __init__(self) -> None:
        super().__init__()

    def forward(self, data: Any) -> Any:
        return data

    def backward(self, data: Any) -> Any:
        return data
