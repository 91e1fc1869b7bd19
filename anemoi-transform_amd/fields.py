"""earthkit-shaped Field / FieldList for the filter path, backed by HBM stacks.

The reference filters see earthkit-data fields through a small surface
(SURVEY.md §8b): ``to_numpy(flatten=, dtype=, index=)``, ``values``, ``shape``,
``grid_points()``, ``to_latlon()``, ``metadata(...)``, ``clone()``; and
FieldLists through iteration, ``len``, indexing, ``metadata(key)``, ``sel``.
earthkit-data is not installed here, so this module supplies look-alikes:

* ``ArrayField``   — a host field built from the ``list-of-dicts`` form the
  reference's tests use (R: tests/conftest.py:40-67): 2-D ``values`` laid out
  latitude-major over distinct ``latitudes`` / ``longitudes``.
* ``DerivedField`` — the three zero-copy wrappers of the reference in one class:
  new data (R: fields.py:157-205 ``NewDataField``), new metadata with the
  override-before-template lookup (R: fields.py:468-568), new lat/lon
  (R: fields.py:318-381).  Its data is either a host array or — the engine's case —
  ``(Stack, level)``: one level of a stack resident in HBM.  ``to_numpy`` is the
  compatibility path and implies a device-to-host copy.
* ``FieldList``    — ``SimpleFieldList`` look-alike (R: fields.py:35-48).
* ``FieldSelection`` — R: fields.py:767-797.

``group_into_stacks`` is how filters see a FieldList: runs of same-grid fields as
whole stacks, so each filter is one kernel launch per stack instead of one numpy
call per field.
"""

from __future__ import annotations

import logging
from typing import Any, Iterable, Iterator

import numpy as np
import torch

from . import native
from . import stack as _stack
from .stack import COLUMNS, Stack

LOG = logging.getLogger(__name__)

MISSING = object()

# R: tests/conftest.py:27 — the keys a "mars" namespace view exposes
MARS_KEYS = ("param", "levelist", "levtype", "type", "step", "date", "time", "number", "expver", "class", "stream", "domain")

_GEO_KEYS = ("latitudes", "longitudes")


class Geography:
    """``metadata().geography`` of a field (R: fields.py:208-315)."""

    def __init__(self, owner: "Field") -> None:
        self._owner = owner

    def latitudes(self, dtype=None) -> np.ndarray:
        lat = self._owner.grid_points()[0]
        return lat if dtype is None else lat.astype(dtype)

    def longitudes(self, dtype=None) -> np.ndarray:
        lon = self._owner.grid_points()[1]
        return lon if dtype is None else lon.astype(dtype)

    def shape(self) -> tuple[int, ...]:
        return tuple(self._owner.shape)

    def resolution(self) -> str:
        return "unknown"

    def mars_area(self) -> list[float]:
        lat, lon = self._owner.grid_points()
        return [np.amax(lat), np.amin(lon), np.amin(lat), np.amax(lon)]

    def mars_grid(self) -> None:
        return None

    def projection(self) -> None:
        return None


class _MetadataView:
    """What ``field.metadata()`` (no arguments) returns: ``get`` / ``keys`` / ``[]`` / ``geography``."""

    def __init__(self, field: "Field") -> None:
        self._field = field
        self.geography = Geography(field)

    def get(self, key: str, default: Any = None) -> Any:
        try:
            return self._field.metadata(key)
        except KeyError:
            return default

    def keys(self):
        return self._field._metadata_keys()

    def __getitem__(self, key: str) -> Any:
        return self._field.metadata(key)

    def __contains__(self, key: str) -> bool:
        return self.get(key, MISSING) is not MISSING

    def items(self):
        return [(k, self.get(k)) for k in self.keys()]


_TYPED_KEY = {"i": int, "l": int, "int": int, "d": float, "float": float, "s": str, "str": str}


class Field:
    """Common surface of host and device fields."""

    # ---- to be provided by subclasses -----------------------------------------------
    shape: tuple[int, ...]

    def _flat(self) -> np.ndarray:  # a fresh, flattened host copy
        raise NotImplementedError

    def grid_points(self) -> tuple[np.ndarray, np.ndarray]:
        raise NotImplementedError

    def _lookup(self, key: str) -> Any:  # MISSING if absent
        raise NotImplementedError

    def _metadata_keys(self) -> list[str]:
        raise NotImplementedError

    def stack_ref(self) -> tuple[Stack, int] | None:
        """``(stack, level)`` if the data lives in HBM, else ``None``."""
        return None

    def target_range(self) -> tuple[int, int, int] | None:
        """``(lo, hi, n_total)`` if the field holds points ``[lo, hi)`` of a grid of ``n_total`` points — the output of
        a target-sharded ``regrid`` on one rank — else ``None``.  Per-point filters that bring a
        full-grid point mask (``apply_mask(path=...)``, ``glacier_mask``) use its window ``[lo, hi)``."""
        return None

    # ---- reference surface -----------------------------------------------------------
    def to_numpy(self, flatten: bool = False, dtype: Any = None, index: Any = None) -> np.ndarray:
        # R: fields.py:178-202 — astype, flatten, index, in that order; always a copy here
        data = self._flat()
        if dtype is not None:
            data = data.astype(dtype)
        if not flatten:
            data = data.reshape(self.shape)
        if index is not None:
            data = data[index]
        return data

    @property
    def values(self) -> np.ndarray:
        return self.to_numpy(flatten=True)

    def to_latlon(self, flatten: bool = True) -> dict[str, np.ndarray]:
        lat, lon = self.grid_points()
        if not flatten:
            lat, lon = lat.reshape(self.shape), lon.reshape(self.shape)
        return dict(lat=lat, lon=lon)

    def metadata(self, *keys: str, namespace: str | None = None, default: Any = MISSING, **_: Any) -> Any:
        if namespace:
            assert len(keys) == 0, (keys, namespace)
            return self._namespace(namespace)
        if len(keys) == 0:
            return _MetadataView(self)
        if len(keys) == 1:  # the common call (`field.metadata("param")`): no closure, no list
            value = self._lookup(keys[0])
            if value is not MISSING:
                return value

        def one(key: str) -> Any:
            value = self._lookup(key)
            if value is MISSING and ":" in key:  # eccodes typed access "levelist:d" (R: rename.py:113-115)
                base, kind = key.rsplit(":", 1)
                cast = _TYPED_KEY.get(kind)
                value = self._lookup(base) if cast is not None else MISSING
                if value is not MISSING:
                    return cast(value)
            if value is MISSING:
                if default is not MISSING:
                    return default
                raise KeyError(key)
            return value

        result = [one(k) for k in keys]
        return result[0] if len(result) == 1 else tuple(result)

    def _namespace(self, namespace: str) -> dict[str, Any]:
        if namespace != "mars":
            return {}
        out = {}
        for key in MARS_KEYS:
            value = self._lookup(key)
            if value is not MISSING:
                out[key] = value
        return out

    def clone(self, *, values: Any = None, **metadata: Any) -> "DerivedField":
        # R: fields.py:131-144 / 600-641: new metadata, same data; earthkit's ``clone(values=...)`` replaces the data
        # (used by user functions of the lambda filter, R: tests/field_filters/test_lambda.py:37)
        if values is not None:
            return DerivedField(self, data=np.asarray(values).reshape(self.shape), metadata=metadata)
        return DerivedField(self, metadata=metadata)

    def __iter__(self):
        raise NotImplementedError(f"{self}: iterating is not supported")  # R: fields.py:146-154


class ArrayField(Field):
    """A host field: array + metadata dict + per-point coordinates."""

    def __init__(self, values: Any, metadata: dict[str, Any], latitudes: np.ndarray, longitudes: np.ndarray,
                 mars: bool = False) -> None:
        self._values = np.asarray(values)
        self._md = dict(metadata)
        # plain list-of-dicts fields expose no "mars" namespace; the reference's MarsUserMetadata
        # test fixture (R: tests/conftest.py:27-38) does — `mars=True` reproduces it
        self._mars = mars
        self._latitudes = np.asarray(latitudes, dtype=np.float64)
        self._longitudes = np.asarray(longitudes, dtype=np.float64)
        self.shape = tuple(self._values.shape)
        assert self._latitudes.shape == self._longitudes.shape == (self._values.size,), (
            self._latitudes.shape, self._longitudes.shape, self._values.shape,
        )

    @classmethod
    def from_dict(cls, spec: dict[str, Any], mars: bool = False) -> "ArrayField":
        """The ``list-of-dicts`` entry form of the reference tests (R: tests/conftest.py:63).

        ``values`` is ``[nlat, nlon]`` over DISTINCT ``latitudes`` / ``longitudes``
        (flattened latitude-major, R: tests/field_filters/test_remove_nans.py:17-45),
        or 1-D with one latitude / longitude per point.
        """
        spec = dict(spec)
        values = np.asarray(spec.pop("values"))
        if values.dtype.kind != "f":
            values = values.astype(np.float64)
        lat = np.asarray(spec["latitudes"], dtype=np.float64)
        lon = np.asarray(spec["longitudes"], dtype=np.float64)
        if values.ndim == 2 and lat.ndim == 1 and lon.ndim == 1 and values.shape == (len(lat), len(lon)):
            lat2, lon2 = np.meshgrid(lat, lon, indexing="ij")
            lat, lon = lat2.reshape(-1), lon2.reshape(-1)
        else:
            lat, lon = lat.reshape(-1), lon.reshape(-1)
        return cls(values, spec, lat, lon, mars=mars)

    def _namespace(self, namespace: str) -> dict[str, Any]:
        if not self._mars:
            return {}
        return super()._namespace(namespace)

    def _flat(self) -> np.ndarray:
        return self._values.flatten()

    def grid_points(self) -> tuple[np.ndarray, np.ndarray]:
        return self._latitudes, self._longitudes

    def _lookup(self, key: str) -> Any:
        return self._md.get(key, MISSING)

    def _metadata_keys(self) -> list[str]:
        return list(self._md.keys())

    def __repr__(self) -> str:
        return f"ArrayField({self._md.get('param')}, shape={self.shape})"


class DerivedField(Field):
    """A field that re-labels a template: new data and / or metadata and / or grid."""

    def __init__(
        self,
        template: Field,
        *,
        data: np.ndarray | None = None,
        stack_level: tuple[Stack, int] | None = None,
        latitudes: np.ndarray | None = None,
        longitudes: np.ndarray | None = None,
        metadata: dict[str, Any] | None = None,
        target_range: tuple[int, int, int] | None = None,
    ) -> None:
        # metadata overrides travel as a dict: keys such as "level" or "data" are legal metadata names
        self._target_range = target_range
        assert data is None or stack_level is None
        stack, level = stack_level if stack_level is not None else (None, None)
        self._template = template
        self._data = data
        self._stack = stack
        self._level = level
        self._latitudes = latitudes
        self._longitudes = longitudes
        self._overrides = dict(metadata or {})
        if data is not None:
            self.shape = tuple(np.shape(data))  # R: fields.py:168-171
        elif stack is not None:
            tshape = tuple(getattr(template, "shape", ()))
            keep = latitudes is None and int(np.prod(tshape or (0,))) == stack.n_pts
            self.shape = tshape if keep else (stack.n_pts,)
        elif latitudes is not None:
            self.shape = (len(latitudes),)
        else:
            self.shape = tuple(template.shape)

    # ---- data ------------------------------------------------------------------------
    def stack_ref(self) -> tuple[Stack, int] | None:
        if self._stack is not None:
            return self._stack, self._level
        if self._data is None:
            return self._template.stack_ref()
        return None

    def target_range(self) -> tuple[int, int, int] | None:
        if self._target_range is not None:
            return self._target_range
        if self._latitudes is None and isinstance(self._template, Field):  # same grid as the template: same window
            inherited = self._template.target_range()
            if inherited is not None and inherited[1] - inherited[0] == int(np.prod(self.shape)):
                return inherited
        return None

    def _flat(self) -> np.ndarray:
        if self._stack is not None:
            return host_level(self._stack, self._level)
        if self._data is not None:
            return np.asarray(self._data).flatten()
        return self._template.to_numpy(flatten=True)

    # ---- grid ------------------------------------------------------------------------
    def grid_points(self) -> tuple[np.ndarray, np.ndarray]:
        if self._latitudes is not None:
            return self._latitudes, self._longitudes
        return self._template.grid_points()

    # ---- metadata: overrides are consulted before the template (R: fields.py:532-546) ----
    def _lookup(self, key: str) -> Any:
        if key in self._overrides:
            value = self._overrides[key]
            if callable(value):
                return value(self, key, self._template.metadata())
            return value
        if self._latitudes is not None and key in _GEO_KEYS:
            return self._latitudes if key == "latitudes" else self._longitudes
        return self._template._lookup(key) if isinstance(self._template, Field) else _foreign_lookup(self._template, key)

    def _metadata_keys(self) -> list[str]:
        # the template's keys only (R: fields.py:508-509; hence the xfails at tests/test_fields.py:45-46)
        if isinstance(self._template, Field):
            return self._template._metadata_keys()
        return list(self._template.metadata().keys())

    def _namespace(self, namespace: str) -> dict[str, Any]:
        # copy of the template's namespace; overrides only for keys already present (R: fields.py:523-530)
        base = dict(self._template.metadata(namespace=namespace))
        for key in list(base.keys()):
            if key in self._overrides:
                base[key] = self._overrides[key]
        return base

    def __getattr__(self, name: str) -> Any:
        # R: fields.py:69-109 — attributes this wrapper does not define are forwarded to the wrapped field, with a warning;
        # `copy` is refused (and so is `clone` there: here every field defines it)
        if name.startswith("_") or name in ("clone", "copy"):
            raise AttributeError(f"{type(self).__name__}: forwarding of `{name}` is not supported")
        LOG.warning("%s: forwarding `%s`", type(self).__name__, name)
        return getattr(self._template, name)

    def __repr__(self) -> str:
        where = f"hbm level {self._level}" if self._stack is not None else ("host array" if self._data is not None else "template data")
        return f"DerivedField({self._template!r}, {where}, metadata={self._overrides})"


def _foreign_lookup(field: Any, key: str) -> Any:
    """Metadata of a real earthkit field used as a template."""
    try:
        return field.metadata(key)
    except KeyError:
        return MISSING


# ---- host copies of device levels ------------------------------------------------------
_HOST_CACHE_ATTR = "_host_fields"


def host_level(stack: Stack, level: int) -> np.ndarray:
    """One level of a stack as a fresh host array (``to_numpy`` returns a copy, R: fields.py:198-199).

    The first access brings the whole stack over in ONE device-to-host copy (field-major, through the
    relayout kernel, into pinned memory) and splits it into one private array per level with a few
    threads; each of those is handed out once, later accesses of the same level copy from the pinned image.
    Stacks are immutable once published to a FieldList, so the image cannot go stale.
    """
    cached = _host_cache.get(id(stack))
    if cached is None or cached[0] is not stack:
        image = stack.numpy()
        if image.nbytes >= _stack._PINNED_MIN_BYTES and image.nbytes <= _SPLIT_MAX_BYTES:
            handouts = list(_stack.copy_pool().map(np.copy, image))
        else:
            handouts = [None] * len(image)
        cached = (stack, image, handouts)
        _host_cache.clear()  # keep at most one stack on the host
        _host_cache[id(stack)] = cached
    ready = cached[2][level]
    if ready is not None:
        cached[2][level] = None
        return ready
    return cached[1][level].copy()


_SPLIT_MAX_BYTES = 8 << 30
_host_cache: dict[int, tuple[Stack, np.ndarray, list]] = {}


# ---- reference factory functions --------------------------------------------------------
def new_field_from_numpy(array: np.ndarray, *, template: Field, **metadata: Any) -> DerivedField:
    """R: fields.py:645-662."""
    return DerivedField(template, data=array, metadata=metadata)


def new_field_from_latitudes_longitudes(template: Field, latitudes: np.ndarray, longitudes: np.ndarray) -> DerivedField:
    """R: fields.py:719-738."""
    return DerivedField(template, latitudes=np.asarray(latitudes), longitudes=np.asarray(longitudes))


def new_field_with_metadata(template: Field, **metadata: Any) -> DerivedField:
    """R: fields.py:683-698."""
    return DerivedField(template, metadata=metadata)


def new_field_with_units(template: Field, units: str) -> DerivedField:
    """R: fields.py:701-716."""
    return DerivedField(template, metadata={"units": units})


def new_field_from_grid(template: Field, grid: Any) -> DerivedField:
    """R: fields.py:741-759 — ``grid`` is any object whose ``latlon()`` gives ``(latitudes, longitudes)`` (R: fields.py:399-407)."""
    latitudes, longitudes = grid.latlon()
    return DerivedField(template, latitudes=np.asarray(latitudes), longitudes=np.asarray(longitudes))


def to_datetime(value: Any) -> "datetime.datetime":
    """ISO string / date / datetime -> datetime (what ``earthkit.data.utils.dates.to_datetime`` is used for here)."""
    import datetime

    if isinstance(value, datetime.datetime):
        return value
    if isinstance(value, datetime.date):
        return datetime.datetime(value.year, value.month, value.day)
    text = str(value)
    if text.endswith("Z"):
        text = text[:-1] + "+00:00"
    return datetime.datetime.fromisoformat(text)


def new_field_with_valid_datetime(template: Field, date: Any) -> DerivedField:
    """R: fields.py:580-597, 665-680 — the field re-dated to ``date`` with ``step`` 0."""
    date = to_datetime(date)
    return DerivedField(template, metadata=dict(date=int(date.strftime("%Y%m%d")), time=int(date.strftime("%H%M")), step=0,
                                                 valid_datetime=date.isoformat()))


def new_field_from_stack(stack: Stack, level: int, *, template: Field, latitudes=None, longitudes=None,
                         metadata: dict[str, Any] | None = None, target_range: tuple[int, int, int] | None = None) -> DerivedField:
    """Engine-side factory: the field is level ``level`` of an HBM stack."""
    return DerivedField(template, stack_level=(stack, level), latitudes=latitudes, longitudes=longitudes, metadata=metadata,
                        target_range=target_range)


class FieldList:
    """Ordered collection of fields (``SimpleFieldList`` look-alike, R: fields.py:35-48)."""

    def __init__(self, fields: Iterable[Any] | None = None) -> None:
        self._fields = list(fields or [])

    def __iter__(self) -> Iterator[Any]:
        return iter(self._fields)

    def __len__(self) -> int:
        return len(self._fields)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return FieldList(self._fields[i])
        return self._fields[i]

    def append(self, field: Any) -> None:
        self._fields.append(field)

    def metadata(self, *keys: str, **kwargs: Any) -> list[Any]:
        return [f.metadata(*keys, **kwargs) for f in self._fields]

    def sel(self, **selection: Any) -> "FieldList":
        def ok(f):
            for key, want in selection.items():
                want = want if isinstance(want, (list, tuple)) else (want,)
                try:
                    if f.metadata(key) not in want:
                        return False
                except KeyError:
                    return False
            return True

        return FieldList([f for f in self._fields if ok(f)])

    def to_numpy(self, **kwargs: Any) -> np.ndarray:
        return np.stack([f.to_numpy(**kwargs) for f in self._fields])

    @property
    def ls(self) -> str:
        return "\n".join(repr(f) for f in self._fields)

    def __repr__(self) -> str:
        return f"FieldList({len(self._fields)} fields)"


def new_fieldlist_from_list(fields: list[Any]) -> FieldList:
    return FieldList(fields)


def new_empty_fieldlist() -> FieldList:
    return FieldList([])


def fieldlist_from_dicts(specs: list[dict[str, Any]], mars: bool = False) -> FieldList:
    """``ekd.from_source("list-of-dicts", specs)`` look-alike (``mars=True``: fields with a MARS namespace,
    as built by R: tests/conftest.py:70-80 ``mars_test_source``)."""
    return FieldList([ArrayField.from_dict(s, mars=mars) for s in specs])


class FieldSelection:
    """Which fields a filter transforms (R: fields.py:767-797)."""

    ALLOWED_KEYS = {"param", "levelist"}

    def __init__(self, **kwargs: Any) -> None:
        self._spec = kwargs
        if not set(self._spec).issubset(self.ALLOWED_KEYS):
            raise ValueError(f"Invalid keys in spec: {tuple(self._spec)} - only {self.ALLOWED_KEYS} are allowed.")
        for key, value in list(self._spec.items()):
            if isinstance(value, (str, int, float, bool)):
                self._spec[key] = (value,)
            elif value is None or (isinstance(value, (list, tuple)) and len(value) == 0):
                del self._spec[key]
            elif not isinstance(value, (list, tuple)):
                raise ValueError(f"Invalid value for key {key}: {value}")
        self._all = len(self._spec) == 0

    def match(self, field: Any) -> bool:
        if self._all:
            return True
        lookup = getattr(field, "_lookup", None)
        if lookup is not None:  # this package's fields: the metadata table directly (a pipeline asks this per field and stage)
            for key, values in self._spec.items():
                value = lookup(key)
                if value is MISSING or value not in values:
                    return False
            return True
        try:
            return all(field.metadata(key) in values for key, values in self._spec.items())
        except KeyError:
            return False


# =================================================================================
# FieldList -> stacks
# =================================================================================
class StackGroup:
    """Fields of a FieldList that share a grid, as one HBM stack.

    ``positions[i]`` is the index in the FieldList of ``fields[i]``, which is stored at level
    ``levels[i]`` of ``stack`` (``levels == range(n)`` unless the group is a sparse view of a bigger stack).
    """

    __slots__ = ("stack", "positions", "fields", "levels")

    def __init__(self, stack: Stack, positions: list[int], fields: list[Any], levels: list[int] | None = None) -> None:
        self.stack = stack
        self.positions = positions
        self.fields = fields
        self.levels = list(range(len(fields))) if levels is None else levels


_upload_dtype: torch.dtype | None = None


def set_upload_dtype(dtype: torch.dtype | None) -> None:
    """Force the dtype host fields are uploaded in (``None``: keep float32 inputs, else float64 —
    the reference's ``to_numpy`` default)."""
    global _upload_dtype
    _upload_dtype = dtype


def _host_dtype(arrays: list[np.ndarray]) -> torch.dtype:
    if _upload_dtype is not None:
        return _upload_dtype
    return torch.float32 if all(a.dtype == np.float32 for a in arrays) else torch.float64


def host_values(field: Any) -> np.ndarray:
    """The flattened values of a host field for upload — without the copy ``to_numpy(flatten=True)`` makes
    (R: fields.py:198-199) when the field is one of ours; the array is only read."""
    if isinstance(field, ArrayField):
        return field._values.reshape(-1)
    return np.asarray(field.to_numpy(flatten=True))


def select_levels(stack: Stack, levels: list[int]) -> Stack:
    """A new stack holding the given levels of ``stack`` (device-side level gather, ``atx_select_levels``)."""
    if levels == list(range(stack.n_lev)):
        return stack
    out = Stack.empty(stack.n_pts, len(levels), stack.dtype, stack.device, stack.layout, zero=True)
    native.select_levels(stack.data, out.data, levels, n_pts=stack.n_pts, n_src_lev=stack.n_lev, src_pitch=stack.pitch,
                         dst_pitch=out.pitch, layout=stack.layout)
    return out


MAX_STACK_LEVELS = 512


def group_into_stacks(fields: Iterable[Any], positions: list[int] | None = None, sparse_ok: bool = False) -> list[StackGroup]:
    """Partition fields into stacks: device fields by the stack they live in, host fields by grid size.

    Host groups are uploaded once (one H2D copy + one relayout launch); device groups
    that cover a whole stack in order are used in place, others are compacted on the
    device — unless ``sparse_ok`` and the group covers at least half of its stack: then the stack
    itself is returned with ``levels`` naming the members (a per-level operator can leave the other
    levels alone, which is cheaper than copying most of the stack first).  ``positions`` restricts
    the grouping to a subset of the list.
    """
    fields = list(fields)
    wanted = range(len(fields)) if positions is None else positions
    buckets: dict[Any, list[int]] = {}
    host_arrays: dict[int, np.ndarray] = {}
    for i in wanted:
        ref = fields[i].stack_ref() if isinstance(fields[i], Field) else None
        if ref is not None:
            key = ("hbm", id(ref[0]))
        else:
            # the width of a field is its OWN (float32 stays float32, everything else is float64 — the dtype rule of DESIGN.md §4):
            # float32 and float64 fields of one grid go into two stacks, so that what a field comes out as never depends on which
            # other fields a filter happened to select with it (fused or filter by filter)
            host_arrays[i] = host_values(fields[i])
            narrow = _upload_dtype is None and host_arrays[i].dtype == np.float32
            key = ("host", int(host_arrays[i].size), narrow)
        buckets.setdefault(key, []).append(i)

    groups = []
    for key, members in buckets.items():
        group_fields = [fields[i] for i in members]
        if key[0] == "hbm":
            stack = group_fields[0].stack_ref()[0]
            levels = [f.stack_ref()[1] for f in group_fields]
            if sparse_ok and levels != list(range(stack.n_lev)) and len(set(levels)) == len(levels) and 2 * len(levels) >= stack.n_lev:
                groups.append(StackGroup(stack, members, group_fields, levels))
            else:
                groups.append(StackGroup(select_levels(stack, levels), members, group_fields))
        else:
            members, group_fields = _variables_together(members, group_fields)
            arrays = [host_arrays[i] for i in members]
            dtype = _host_dtype(arrays)
            # long lists (config 4: thousands of fields on one grid) become several stacks of at most MAX_STACK_LEVELS:
            # the operator tables of an 8-stage program then always fit the kernels' shared memory, the staging chunks
            # stay bounded, and stacks that share a plan still go through one batched launch
            for first in range(0, len(members), MAX_STACK_LEVELS):
                part = slice(first, first + MAX_STACK_LEVELS)
                groups.append(StackGroup(Stack.from_fields(arrays[part], dtype=dtype, dev=_stack.device()), members[part], group_fields[part]))
    return groups


def _variables_together(members: list[int], group_fields: list[Any]) -> tuple[list[int], list[Any]]:
    """The fields of one host group with those of one ``param`` next to each other, otherwise in the order they came.

    Which level of the stack a field becomes is internal (results go back to their positions in the list), so a list that comes
    level by level — t, q, t, q, ... — is uploaded as "all levels of t, then all of q": a filter that treats the variables
    differently (``convert`` on t, ``clip`` on q) then has a program of RUNS of levels, which the kernels take by value, instead of
    one that changes at every level (per-level tables)."""
    def name(f):
        try:
            return str(f.metadata("param"))
        except Exception:  # noqa: BLE001 - fields without a param stay where they are
            return ""

    names = [name(f) for f in group_fields]
    changes = sum(a != b for a, b in zip(names, names[1:]))
    if changes < len(set(names)):  # already one run per variable
        return members, group_fields
    first_seen = {n: i for i, n in reversed(list(enumerate(names)))}
    order = sorted(range(len(names)), key=lambda j: (first_seen[names[j]], j))
    return [members[j] for j in order], [group_fields[j] for j in order]


def fields_to_stack(fields: list[Any]) -> Stack:
    """One HBM stack whose level ``i`` is ``fields[i]`` (all on the same grid), in that order.

    Fields that already are levels of one stack are selected on the device (``atx_select_levels``); host fields
    are uploaded as one staged stack; levels of several stacks are gathered stack by stack.
    """
    refs = [f.stack_ref() if isinstance(f, Field) else None for f in fields]
    if all(r is not None for r in refs) and all(r[0] is refs[0][0] for r in refs):
        return select_levels(refs[0][0], [r[1] for r in refs])
    # mixed origins: host fields go up as one staged stack, every device stack contributes its levels by one level gather
    dev = _stack.device()
    host_pos = [i for i, r in enumerate(refs) if r is None]
    host_arrays = [host_values(fields[i]) for i in host_pos]
    host_arrays = [a if a.dtype in (np.float32, np.float64) else a.astype(np.float64) for a in host_arrays]
    device_stacks: dict[int, Stack] = {id(r[0]): r[0] for r in refs if r is not None}
    if _upload_dtype is not None:
        dtype = _upload_dtype
    else:
        all_f32 = all(a.dtype == np.float32 for a in host_arrays) and all(st.dtype == torch.float32 for st in device_stacks.values())
        dtype = torch.float32 if all_f32 else torch.float64
    sizes = {a.size for a in host_arrays} | {st.n_pts for st in device_stacks.values()}
    assert len(sizes) == 1, "fields of one stack must share a grid"
    n_pts = sizes.pop()
    if not device_stacks:
        return Stack.from_fields(host_arrays, dtype=dtype, dev=dev)
    out = Stack.empty(n_pts, len(fields), dtype, dev, COLUMNS, zero=True)
    sources: list[tuple[Stack, dict[int, int]]] = []  # (stack, {destination level: source level})
    if host_pos:
        staged = Stack.from_fields(host_arrays, dtype=dtype, dev=dev)
        sources.append((staged, {pos: j for j, pos in enumerate(host_pos)}))
    for key, st in device_stacks.items():
        if st.dtype != dtype:  # mixed precision across stacks: widen the narrower one (a cast, not arithmetic)
            st = Stack(st.data.to(dtype), st.n_pts, st.n_lev, st.layout)
        sources.append((st.to_layout(COLUMNS), {i: r[1] for i, r in enumerate(refs) if r is not None and id(r[0]) == key}))
    for st, wanted in sources:
        level_map = [wanted.get(j, -1) for j in range(len(fields))]
        native.select_levels(st.data, out.data, level_map, n_pts=n_pts, n_src_lev=st.n_lev, src_pitch=st.pitch,
                             dst_pitch=out.pitch, layout=COLUMNS)
    return out
