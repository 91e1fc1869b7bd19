"""The filter plugin registry of the hot path (mirror of R: filters/__init__.py:19-64).

Registered names:
  field filters   regrid, apply_mask_fields, remove_nans_fields, rescale, convert,
                  orog_to_z_fields, z_to_orog_fields, clip_fields, impute_nans_fields,
                  lnsp_to_sp, sp_to_lnsp, glacier_mask, noop,
                  snow_depth_m, snow_cover, cos_sin_from_rad, cos_sin_mean_wave_direction,
                  w_to_wz, wz_to_w, uv_to_ddff, ddff_to_uv, sum, accum_to_interval   (multi-input, filters/multi.py)
                  rodeo_opera_clipping, rodeo_opera_preprocessing, oras6_clipping, land_parameters,
                  r_to_d, d_to_r, q_to_r, r_to_q, q_to_r_height_with_p, r_to_q_height_with_p
                                                             (multi-input, filters/domain.py)
                  rename_fields, clear_step, repeat_members, earthkitfieldlambda, empty,
                  icon_refinement_level   (re-labelling / re-listing, filters/metadata.py)
  dispatchers     mask (alias apply_mask), remove_nans (alias drop_nans),
                  geopotential_to_height (alias orog_to_z), height_to_geopotential (alias z_to_orog),
                  clip (alias clipper), impute_nans (alias replace_nans), rename

The dispatchers pick the field filter from the configuration keys exactly as the
reference does; configurations that select the reference's *tabular* (pandas)
filters raise ``NotImplementedError`` — observations are outside this path
(SURVEY.md §2.1 row 16).
"""

from __future__ import annotations

from typing import Any

from ..core import DispatchingFilter, Filter, filter_registry

# importing the modules registers the field filters
from . import masks as _masks  # noqa: E402
from . import domain as _domain  # noqa: E402
from . import metadata as _metadata  # noqa: E402
from . import multi as _multi  # noqa: E402
from . import pointwise as _pointwise  # noqa: E402
from . import regrid as _regrid  # noqa: E402
from .masks import MaskVariable, RemoveNaNs as RemoveNaNsFields
from .pointwise import Clipper, ImputeNaNs as ImputeNaNsFields, Orography


def _tabular(name: str) -> NotImplementedError:
    return NotImplementedError(
        f"'{name}': this configuration selects the tabular (pandas DataFrame) filter of the reference, "
        "which is outside the gridded-field hot path this package implements"
    )


class Mask(DispatchingFilter):
    """R: filters/mask.py:19-35."""

    def __init__(self, **config: Any) -> None:
        if "path" in config or "mask_param" in config:
            self.filter = MaskVariable(**config)
        else:
            raise _tabular("mask")

    def forward_fields(self, data: Any) -> Any:
        return self.filter.forward(data)


class RemoveNaNs(DispatchingFilter):
    """R: filters/remove_nans.py:19-47."""

    def __init__(self, **config: Any) -> None:
        if len(config) == 0:
            self.field_filter = RemoveNaNsFields()
        elif ("columns" in config) or ("column_prefix" in config) or ("how" in config):
            self.field_filter = None
        else:
            self.field_filter = RemoveNaNsFields(**config)

    def forward_fields(self, data: Any) -> Any:
        if self.field_filter is None:
            raise ValueError("Ambigious config for RemoveNaNs filter.")
        return self.field_filter.forward(data)


class GeopotentialToHeight(DispatchingFilter):
    """R: filters/geopotential_to_height.py:19-47."""

    def __init__(self, **config: Any) -> None:
        config["geopotential"] = config.get("geopotential", "z")
        if ("height" in config) and ("orography" in config):
            raise ValueError("Must specify either 'height' or 'orography' parameter, but not both.")
        if "height" not in config:
            config["height"] = config.pop("orography", "orog")
        self.field_filter = Orography(geopotential=config["geopotential"], orography=config["height"])

    def forward_fields(self, data: Any) -> Any:
        return self.field_filter.forward(data)

    def backward_fields(self, data: Any) -> Any:
        return self.field_filter.backward(data)

    def patch_data_request(self, data_request: dict) -> dict:
        return self.field_filter.patch_data_request(data_request)


class Clip(DispatchingFilter):
    """R: filters/clip.py:19-35."""

    def __init__(self, **config: Any) -> None:
        if "param" in config and isinstance(config["param"], str):
            self.filter = Clipper(**config)
        else:
            raise _tabular("clip")

    def forward_fields(self, data: Any) -> Any:
        return self.filter.forward(data)


class ImputeNaNs(DispatchingFilter):
    """R: filters/impute_nans.py:19-49."""

    def __init__(self, **config: Any) -> None:
        if len(config) == 0:
            self.field_filter = ImputeNaNsFields()  # raises like the reference: required inputs are missing
        elif ("columns" in config) or ("column_prefix" in config):
            self.field_filter = None
        else:
            self.field_filter = ImputeNaNsFields(**config)

    def forward_fields(self, data: Any) -> Any:
        if self.field_filter is None:
            raise ValueError("Ambiguous config for ImputeNaNs field filter.")
        return self.field_filter.forward(data)


filter_registry.register("mask", Mask, aliases=["apply_mask"])
filter_registry.register("remove_nans", RemoveNaNs, aliases=["drop_nans"])
filter_registry.register("geopotential_to_height", GeopotentialToHeight, aliases=["orog_to_z"])
filter_registry.register("height_to_geopotential", GeopotentialToHeight.reversed, aliases=["z_to_orog"])
filter_registry.register("clip", Clip, aliases=["clipper"])
filter_registry.register("impute_nans", ImputeNaNs, aliases=["replace_nans"])


def create_filter_by_name(name: str, *, context: Any = None, **config: Any) -> Filter:
    """R: filters/__init__.py:36-40."""
    filter = filter_registry.create(name, **config)
    filter.context = context
    return filter


def create_filter(context: Any, config: Any) -> Filter:
    """R: filters/__init__.py:43-60 — ``config`` is ``"name"`` or ``{"name": {kwargs}}``."""
    filter = filter_registry.from_config(config)
    filter.context = context
    return filter


__all__ = ["filter_registry", "create_filter", "create_filter_by_name"]
