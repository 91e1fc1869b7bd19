"""Domain filters of the reference's MatchingFieldsFilter family.

Numpy-only in the reference (no third-party arithmetic behind them), bit for bit here:

  rodeo_opera_clipping        R: filters/fields/rodeo_opera_clipping.py
  rodeo_opera_preprocessing   R: filters/fields/rodeo_opera_preprocessing.py
  oras6_clipping              R: filters/fields/oras6_clipping.py
  land_parameters             R: filters/fields/land_parameters.py

Humidity conversions whose arithmetic is earthkit-meteo's (restated from its published formulas, pinned by the reference's test
vectors at np.allclose — second half of this file):

  r_to_d / d_to_r                                 R: filters/fields/dewpoint.py
  q_to_r / r_to_q                                 R: filters/fields/q_to_r.py
  q_to_r_height_with_p / r_to_q_height_with_p     R: filters/fields/q_height.py:57-152

The first four are ``MatchingFieldsFilter``s there: fields grouped by their MARS key minus ``param``, one numpy expression per group.
Here the OPERA filters run ONE ``atx_combine_stack`` launch for all groups (``StackMatchingFilter``); ``oras6_clipping`` runs one
launch per group over the group's 14 fields (they share ONE ice mask, the group's ``siconc``); ``land_parameters`` runs one table
look-up launch per output variable.
"""

from __future__ import annotations

import logging
from typing import Any, Iterator

import torch

from .. import native
from ..core import filter_registry
from ..fields import fields_to_stack, new_field_from_stack
from .engine import PointMask
from .masks import _level_tensor
from .multi import MatchingFieldsFilter, MatchingSpec, StackMatchingFilter

LOG = logging.getLogger(__name__)

MAX_TP = 10000  # R: rodeo_opera_preprocessing.py:30, rodeo_opera_clipping.py:22


@filter_registry.register("rodeo_opera_clipping")
class RodeoOperaClipping(StackMatchingFilter):
    """OPERA composites: precipitation limited to ``[0, max_total_precipitation]`` and divided by 1000, quality index to ``[0, 1]``
    (R: rodeo_opera_clipping.py:25-103).  The limits are boolean-mask assignments there (``v[v < 0] = 0; v[v >= m] = m``): NaNs stay."""

    MATCHING = MatchingSpec(select="param", forward=("total_precipitation", "quality"))

    def __init__(self, *, total_precipitation: str = "tp", max_total_precipitation: float = MAX_TP, quality: str = "qi") -> None:
        self.total_precipitation = total_precipitation
        self.max_total_precipitation = max_total_precipitation
        self.quality = quality
        super().__init__()

    def forward_plan(self, total_precipitation: Any, quality: Any):
        return (native.COMB_OPERA_CLIP, 0,
                [(total_precipitation, dict(param=self.total_precipitation)), (quality, dict(param=self.quality))],
                float(self.max_total_precipitation))

    def forward_transform(self, total_precipitation: Any = None, quality: Any = None) -> Iterator[Any]:
        return super().forward_transform(total_precipitation=total_precipitation, quality=quality)


@filter_registry.register("rodeo_opera_preprocessing")
class RodeoOperaPreProcessing(StackMatchingFilter):
    """OPERA composites: the data mask ``dm`` applied (1 = no data and 3 = inf: precipitation NaN; 2 = undetected: precipitation and
    quality 0), then the same limits as ``rodeo_opera_clipping`` without the division (R: rodeo_opera_preprocessing.py:100-205).
    The mask field leaves the stream unless ``return_mask``; it then follows the two results of its group, as in the reference."""

    MATCHING = MatchingSpec(select="param", forward=("total_precipitation", "quality", "mask"))

    def __init__(self, *, total_precipitation: str = "tp", quality: str = "qi", mask: str = "dm",
                 max_total_precipitation: float = MAX_TP, return_mask: bool = False) -> None:
        self.total_precipitation = total_precipitation
        self.quality = quality
        self.mask = mask
        self.max_total_precipitation = max_total_precipitation
        self.return_mask = return_mask
        super().__init__()

    def forward_plan(self, total_precipitation: Any, quality: Any, mask: Any):
        return (native.COMB_OPERA_PREPROCESS, 0,
                [(total_precipitation, dict(param=self.total_precipitation)), (quality, dict(param=self.quality))],
                float(self.max_total_precipitation), [mask] if self.return_mask else [])

    def _check_results(self, direction: str, stacks: list[Any]) -> None:
        # R: rodeo_opera_preprocessing.py:91-93 — a warning, per call there, over all groups of one grid here
        if not LOG.isEnabledFor(logging.WARNING):
            return
        tp, qi = (native.reduce_stack(s.data, native.RED_NANCOUNT, n_pts=s.n_pts, n_lev=s.n_lev, pitch=s.pitch, layout=s.layout)
                  for s in stacks)
        if tp != qi:
            LOG.warning(f"Mismatch between NaNs on tp {int(tp)} and qi {int(qi)}")

    def forward_transform(self, total_precipitation: Any = None, quality: Any = None, mask: Any = None) -> Iterator[Any]:
        return super().forward_transform(total_precipitation=total_precipitation, quality=quality, mask=mask)


# what each ORAS6 field is to the cleaning kernel (R: oras6_clipping.py:196-215), in the order the reference yields them (:217-230)
_ORAS6_KINDS = (
    ("siconc", native.ORAS6_KEEP), ("siue", native.ORAS6_ZERO), ("sivn", native.ORAS6_ZERO), ("icesalt", native.ORAS6_ZERO),
    ("sihc", native.ORAS6_HEAT), ("snhc", native.ORAS6_HEAT), ("sipf", native.ORAS6_ZERO), ("sitemptop", native.ORAS6_TEMPERATURE),
    ("sntemp", native.ORAS6_TEMPERATURE), ("snvol", native.ORAS6_ZERO), ("sivol", native.ORAS6_ZERO), ("sialb", native.ORAS6_ZERO),
    ("vasit", native.ORAS6_TEMPERATURE), ("tos", native.ORAS6_SURFACE),
)


@filter_registry.register("oras6_clipping")
class Oras6Clipping(MatchingFieldsFilter):
    """ORAS6 ocean / sea-ice variables cleaned where the ice concentration is at most 1e-5 (R: oras6_clipping.py:24-231): velocities,
    salinity, heat contents, pressure, volumes and albedo become 0 there, the three temperatures 273.15; heat contents >= -1e-5 become 0
    everywhere; the surface temperature is raised to 271.15 - 1e-5; a snow temperature whose maximum is below 100 is taken to be in
    Celsius and shifted by 273.15 first.  One launch per group over its 14 fields."""

    MATCHING = MatchingSpec(
        select="param",
        forward=("siue", "sivn", "siconc", "icesalt", "sihc", "snhc", "sipf", "sitemptop", "sntemp", "snvol", "sivol", "sialb", "vasit",
                 "tos"),
    )

    def __init__(self, *, siue: str = "avg_siue", sivn: str = "avg_sivn", siconc: str = "avg_siconc", icesalt: str = "avg_icesalt",
                 sihc: str = "avg_sihc", snhc: str = "avg_snhc", sipf: str = "avg_sipf", sitemptop: str = "avg_sitemptop",
                 sntemp: str = "avg_sntemp", snvol: str = "avg_snvol", sivol: str = "avg_sivol", sialb: str = "avg_sialb",
                 vasit: str = "avg_vasit", tos: str = "avg_tos") -> None:
        vars(self).update(siue=siue, sivn=sivn, siconc=siconc, icesalt=icesalt, sihc=sihc, snhc=snhc, sipf=sipf, sitemptop=sitemptop,
                          sntemp=sntemp, snvol=snvol, sivol=sivol, sialb=sialb, vasit=vasit, tos=tos)
        super().__init__()

    def forward_transform(self, siue: Any, sivn: Any, siconc: Any, icesalt: Any, sihc: Any, snhc: Any, sipf: Any, sitemptop: Any,
                          sntemp: Any, snvol: Any, sivol: Any, sialb: Any, vasit: Any, tos: Any) -> Iterator[Any]:
        given = dict(siue=siue, sivn=sivn, siconc=siconc, icesalt=icesalt, sihc=sihc, snhc=snhc, sipf=sipf, sitemptop=sitemptop,
                     sntemp=sntemp, snvol=snvol, sivol=sivol, sialb=sialb, vasit=vasit, tos=tos)
        kinds = dict(_ORAS6_KINDS)
        # R: oras6_clipping.py:190-191 `np.nanmax(sntemp) < 100`: no value at or above 100 (NaNs ignored; an all-NaN field is unchanged by
        # the shift either way) — one comparison pass over the one field
        values, stride, n_points = _level_tensor(sntemp)
        if PointMask.build(values, n_points, cmp=native.CMP_GE, threshold=100.0, stride=stride).count() == 0:
            kinds["sntemp"] = native.ORAS6_CELSIUS
        names = [name for name, _ in _ORAS6_KINDS]
        x = fields_to_stack([given[name] for name in names])
        ice = fields_to_stack([siconc])  # one level: the contiguous field
        assert ice.n_lev == 1 and ice.data.numel() == ice.n_pts, "a one-level stack is the field itself, contiguous"
        if ice.dtype != x.dtype:
            ice = type(ice)(ice.data.to(x.dtype), ice.n_pts, ice.n_lev, ice.layout)
        y = x.new_like(zero=False)
        what = torch.tensor([float(kinds[name]) for name in names], dtype=torch.float64, device=x.device)
        native.combine_stack(native.COMB_ORAS6, [x.data, ice.data], [y.data], n_pts=x.n_pts, n_lev=x.n_lev, pitch=x.pitch,
                             layout=x.layout, level_param=what)
        for level, name in enumerate(names):
            yield new_field_from_stack(y, level, template=given[name], metadata=dict(param=getattr(self, name)))


# R: land_parameters.py:20-52 — class -> parameters
SOIL_TYPE_DIC = {
    0: {"theta_pwp": 0.0, "theta_cap": 0.0}, 1: {"theta_pwp": 0.059, "theta_cap": 0.244}, 2: {"theta_pwp": 0.151, "theta_cap": 0.347},
    3: {"theta_pwp": 0.133, "theta_cap": 0.383}, 4: {"theta_pwp": 0.279, "theta_cap": 0.448}, 5: {"theta_pwp": 0.335, "theta_cap": 0.541},
    6: {"theta_pwp": 0.267, "theta_cap": 0.663}, 7: {"theta_pwp": 0.151, "theta_cap": 0.347},
}
VEG_TYPE_DIC = {
    0: {"veg_rsmin": 250.0, "veg_cov": 0.0, "veg_z0m": 0.013}, 1: {"veg_rsmin": 125.0, "veg_cov": 0.9, "veg_z0m": 0.25},
    2: {"veg_rsmin": 80.0, "veg_cov": 0.85, "veg_z0m": 0.1}, 3: {"veg_rsmin": 395.0, "veg_cov": 0.9, "veg_z0m": 2.0},
    4: {"veg_rsmin": 320.0, "veg_cov": 0.9, "veg_z0m": 2.0}, 5: {"veg_rsmin": 215.0, "veg_cov": 0.9, "veg_z0m": 2.0},
    6: {"veg_rsmin": 320.0, "veg_cov": 0.99, "veg_z0m": 2.0}, 7: {"veg_rsmin": 100.0, "veg_cov": 0.7, "veg_z0m": 0.5},
    8: {"veg_rsmin": 250.0, "veg_cov": 0.0, "veg_z0m": 0.013}, 9: {"veg_rsmin": 45.0, "veg_cov": 0.5, "veg_z0m": 0.03},
    10: {"veg_rsmin": 110.0, "veg_cov": 0.9, "veg_z0m": 0.5}, 11: {"veg_rsmin": 45.0, "veg_cov": 0.1, "veg_z0m": 0.03},
    12: {"veg_rsmin": 0.0, "veg_cov": 0.0, "veg_z0m": 0.0013}, 13: {"veg_rsmin": 130.0, "veg_cov": 0.6, "veg_z0m": 0.25},
    14: {"veg_rsmin": 0.0, "veg_cov": 0.0, "veg_z0m": 0.0001}, 15: {"veg_rsmin": 0.0, "veg_cov": 0.0, "veg_z0m": 0.0001},
    16: {"veg_rsmin": 230.0, "veg_cov": 0.5, "veg_z0m": 0.5}, 17: {"veg_rsmin": 110.0, "veg_cov": 0.4, "veg_z0m": 0.1},
    18: {"veg_rsmin": 180.0, "veg_cov": 0.9, "veg_z0m": 1.50}, 19: {"veg_rsmin": 175.0, "veg_cov": 0.9, "veg_z0m": 1.1},
    20: {"veg_rsmin": 150.0, "veg_cov": 0.6, "veg_z0m": 0.02},
}


def read_crosswalking_table(classes: Any, param_dic: dict[int, dict[str, float]]) -> list[Any]:
    """One float64 HBM stack per key of the table, ``param_dic[class][key]`` for every point of the class stack
    (R: land_parameters.py:55-72).  A class that is not a key of the table raises KeyError, as there."""
    if classes.dtype != torch.float64:  # the reference's `np.array([...])` of Python floats is float64 whatever the classes were
        classes = type(classes)(classes.data.to(torch.float64), classes.n_pts, classes.n_lev, classes.layout)
    n = len(param_dic)
    assert sorted(param_dic) == list(range(n)), "the classes of a crosswalking table are 0 .. n-1"
    shape = dict(n_pts=classes.n_pts, n_lev=classes.n_lev, pitch=classes.pitch, layout=classes.layout)
    out = []
    for key in param_dic[0].keys():
        table = torch.tensor([float(n)] + [float(param_dic[c][key]) for c in range(n)], dtype=torch.float64, device=classes.device)
        values = classes.new_like(zero=False)
        native.combine_stack(native.COMB_LOOKUP, [classes.data], [values.data], level_param=table, **shape)
        if not out and native.reduce_stack(values.data, native.RED_NANCOUNT, **shape) > 0:  # the same classes for every key: checked once
            lo, hi = native.reduce_stack(classes.data, native.RED_MINMAX, **shape)
            raise KeyError(f"class outside the table's 0 .. {n - 1} (classes range from {lo} to {hi}; fractions and NaN are no classes)")
        out.append(values)
    return out


@filter_registry.register("land_parameters")
class LandParameters(MatchingFieldsFilter):
    """Static land parameters looked up from the vegetation and soil classes (R: land_parameters.py:75-146): minimum stomatal
    resistance, cover and roughness length for the high and the low vegetation type, wilting point and field capacity for the soil."""

    MATCHING = MatchingSpec(select="param", forward=("high_veg_type", "low_veg_type", "soil_type"))

    def __init__(self, *, high_veg_type: str = "tvh", low_veg_type: str = "tvl", soil_type: str = "slt", hveg_rsmin: str = "hveg_rsmin",
                 hveg_cov: str = "hveg_cov", hveg_z0m: str = "hveg_z0m", lveg_rsmin: str = "lveg_rsmin", lveg_cov: str = "lveg_cov",
                 lveg_z0m: str = "lveg_z0m", theta_pwp: str = "theta_pwp", theta_cap: str = "theta_cap") -> None:
        vars(self).update(high_veg_type=high_veg_type, low_veg_type=low_veg_type, soil_type=soil_type, hveg_rsmin=hveg_rsmin,
                          hveg_cov=hveg_cov, hveg_z0m=hveg_z0m, lveg_rsmin=lveg_rsmin, lveg_cov=lveg_cov, lveg_z0m=lveg_z0m,
                          theta_pwp=theta_pwp, theta_cap=theta_cap)
        super().__init__()

    def forward_transform(self, high_veg_type: Any, low_veg_type: Any, soil_type: Any) -> Iterator[Any]:
        for template, table, names in (
            (high_veg_type, VEG_TYPE_DIC, (self.hveg_rsmin, self.hveg_cov, self.hveg_z0m)),
            (low_veg_type, VEG_TYPE_DIC, (self.lveg_rsmin, self.lveg_cov, self.lveg_z0m)),
            (soil_type, SOIL_TYPE_DIC, (self.theta_pwp, self.theta_cap)),
        ):
            if len(template.shape) != 1:  # R: land_parameters.py:71 iterates the array: the rows of a 2-D field are no dictionary keys
                raise TypeError(f"unhashable type: 'numpy.ndarray' (land_parameters takes 1-D fields, got shape {tuple(template.shape)})")
            for name, stack in zip(names, read_crosswalking_table(fields_to_stack([template]), table)):
                yield new_field_from_stack(stack, 0, template=template, metadata=dict(param=name))


# =================================================================================
# humidity conversions (the arithmetic is earthkit-meteo's thermo.array, restated in atx_combine.hip from its published form
# and pinned by the reference's test vectors at np.allclose — see oracle.py)
# =================================================================================
HUMIDITY_NOTE = ("arithmetic restated from earthkit-meteo's thermo.array (IFS saturation formulas, mixed phase; package absent here); "
                 "pinned only at the reference's test points (tests/field_filters/test_dewpoint.py, test_pressure_level_humidity.py) at "
                 "np.allclose, not bit for bit")


class DewPoint(StackMatchingFilter):
    """Relative humidity (%) and temperature (K) -> dewpoint temperature (K), and back (R: filters/fields/dewpoint.py:24-76).
    A relative humidity of exactly 0 is replaced by 1e-4 before the dewpoint is taken, as in the reference."""

    MATCHING = MatchingSpec(select="param", forward=("relative_humidity", "temperature"), backward=("dewpoint", "temperature"))
    PARITY_NOTE = HUMIDITY_NOTE

    def __init__(self, *, relative_humidity: str = "r", temperature: str = "t", dewpoint: str = "d", return_inputs: Any = "all") -> None:
        self.return_inputs = return_inputs
        self.relative_humidity = relative_humidity
        self.temperature = temperature
        self.dewpoint = dewpoint
        super().__init__()

    def forward_plan(self, relative_humidity: Any, temperature: Any):
        return native.COMB_R_TO_D, 0, [(relative_humidity, dict(param=self.dewpoint))], None

    def backward_plan(self, dewpoint: Any, temperature: Any):
        return native.COMB_D_TO_R, 0, [(temperature, dict(param=self.relative_humidity))], None

    def forward_transform(self, relative_humidity: Any = None, temperature: Any = None) -> Iterator[Any]:
        return super().forward_transform(relative_humidity=relative_humidity, temperature=temperature)

    def backward_transform(self, dewpoint: Any = None, temperature: Any = None) -> Iterator[Any]:
        return super().backward_transform(dewpoint=dewpoint, temperature=temperature)


filter_registry.register("r_to_d", DewPoint)
filter_registry.register("d_to_r", DewPoint.reversed)


def _pressure_level(field: Any) -> float:
    return float(field.metadata("levelist"))  # hPa; the kernel multiplies by 100 (R: q_to_r.py:72, :78)


class HumidityConversion(StackMatchingFilter):
    """Specific humidity (kg/kg) and temperature on a pressure level -> relative humidity (%), and back
    (R: filters/fields/q_to_r.py:21-84).  The pressure is 100 x the ``levelist`` of the humidity (forward) or temperature (backward) field."""

    MATCHING = MatchingSpec(select="param", forward=("humidity", "temperature"), backward=("relative_humidity", "temperature"))
    PARITY_NOTE = HUMIDITY_NOTE

    def __init__(self, *, relative_humidity: str = "r", temperature: str = "t", humidity: str = "q", return_inputs: Any = "all") -> None:
        self.return_inputs = return_inputs
        self.relative_humidity = relative_humidity
        self.temperature = temperature
        self.humidity = humidity
        super().__init__()

    def forward_plan(self, humidity: Any, temperature: Any):
        return native.COMB_Q_TO_R, 0, [(humidity, dict(param=self.relative_humidity))], _pressure_level(humidity)

    def backward_plan(self, relative_humidity: Any, temperature: Any):
        return native.COMB_R_TO_Q, 0, [(relative_humidity, dict(param=self.humidity))], _pressure_level(temperature)

    def forward_transform(self, humidity: Any = None, temperature: Any = None) -> Iterator[Any]:
        return super().forward_transform(humidity=humidity, temperature=temperature)

    def backward_transform(self, relative_humidity: Any = None, temperature: Any = None) -> Iterator[Any]:
        return super().backward_transform(relative_humidity=relative_humidity, temperature=temperature)


filter_registry.register("q_to_r", HumidityConversion)
filter_registry.register("r_to_q", HumidityConversion.reversed)


class SpecificToRelativeAtHeightLevelWithP(StackMatchingFilter):
    """The same conversion with the pressure given as a third field (Pa), e.g. on height levels
    (R: filters/fields/q_height.py:57-152).  The two other filters of that file (``q_to_r_height``, ``q_to_d_height``) derive the
    pressure from model levels with earthkit-meteo's ``vertical.pressure_at_height_levels`` and are not implemented."""

    MATCHING = MatchingSpec(
        select="param",
        forward=("specific_humidity_at_height_level", "temperature_at_height_level", "pressure_at_height_level"),
        backward=("relative_humidity_at_height_level", "temperature_at_height_level", "pressure_at_height_level"),
        vertical=False,
    )
    PARITY_NOTE = HUMIDITY_NOTE

    def __init__(self, *, specific_humidity_at_height_level: str = "q", relative_humidity_at_height_level: str = "r",
                 pressure_at_height_level: str = "pres", temperature_at_height_level: str = "t",
                 return_inputs: Any = ("specific_humidity_at_height_level", "relative_humidity_at_height_level",
                                       "temperature_at_height_level", "pressure_at_height_level")) -> None:
        self.return_inputs = return_inputs
        self.specific_humidity_at_height_level = specific_humidity_at_height_level
        self.relative_humidity_at_height_level = relative_humidity_at_height_level
        self.temperature_at_height_level = temperature_at_height_level
        self.pressure_at_height_level = pressure_at_height_level
        super().__init__()

    def forward_plan(self, specific_humidity_at_height_level: Any, temperature_at_height_level: Any, pressure_at_height_level: Any):
        return (native.COMB_Q_TO_R, 0, [(specific_humidity_at_height_level, dict(param=self.relative_humidity_at_height_level))], None)

    def backward_plan(self, relative_humidity_at_height_level: Any, temperature_at_height_level: Any, pressure_at_height_level: Any):
        return (native.COMB_R_TO_Q, 0, [(relative_humidity_at_height_level, dict(param=self.specific_humidity_at_height_level))], None)

    def forward_transform(self, specific_humidity_at_height_level: Any = None, temperature_at_height_level: Any = None,
                          pressure_at_height_level: Any = None) -> Iterator[Any]:
        return super().forward_transform(specific_humidity_at_height_level=specific_humidity_at_height_level,
                                         temperature_at_height_level=temperature_at_height_level,
                                         pressure_at_height_level=pressure_at_height_level)

    def backward_transform(self, relative_humidity_at_height_level: Any = None, temperature_at_height_level: Any = None,
                           pressure_at_height_level: Any = None) -> Iterator[Any]:
        return super().backward_transform(relative_humidity_at_height_level=relative_humidity_at_height_level,
                                          temperature_at_height_level=temperature_at_height_level,
                                          pressure_at_height_level=pressure_at_height_level)


filter_registry.register("q_to_r_height_with_p", SpecificToRelativeAtHeightLevelWithP)
filter_registry.register("r_to_q_height_with_p", SpecificToRelativeAtHeightLevelWithP.reversed)
