"""Host-side mirror of the reference's plugin / operator interface for the filter path.

Same names, argument meaning and error behaviour as the reference objects a
caller of the hot path touches, so the parity tests read like the reference's
own tests:

* ``Registry``            — the call surface of ``anemoi.utils.registry.Registry`` that the
  reference uses (R: filters/__init__.py:19-64, filters/fields/__init__.py:13,
  filters/fields/regrid.py:87, filters/fields/orog_to_z.py:97-98, filters/mask.py:35,
  commands/filters.py:45, docs/scripts/list-filters.py:89).  That class lives in the
  third-party anemoi-utils package (not in /root/reference); the surface here is the
  one inferred from those call sites.
* ``Transform`` / ``ReversedTransform`` — R: transform.py:27-244
* ``Filter`` / ``DispatchingFilter`` / ``SingleFieldFilter`` — R: filter.py:29-202
* ``Workflow`` / ``Pipeline`` / ``Source`` — R: workflow.py, workflows/pipeline.py:18-64, source.py

Nothing here computes: arithmetic lives in libatx (``native``); these classes
only route FieldLists.
"""

from __future__ import annotations

import logging
from abc import ABC, ABCMeta, abstractmethod
from typing import Any, Callable

LOG = logging.getLogger(__name__)

# ---- notes a filter logs about its own parity (unpinned third-party arithmetic, a jump in a statement): a pipeline that builds its
#      filters per batch or per date must not repeat them — the first construction in a process says it at the given level, later
#      ones at DEBUG
_NOTES_SAID: set[Any] = set()


def say_once(logger: logging.Logger, key: Any, message: str, *args: Any, level: int = logging.WARNING) -> None:
    logger.log(logging.DEBUG if key in _NOTES_SAID else level, message, *args)
    _NOTES_SAID.add(key)


def reset_notes() -> None:
    """Forget which notes were said (tests)."""
    _NOTES_SAID.clear()


# =================================================================================
# Registry
# =================================================================================
class Registry:
    """Name -> factory table with aliases, ``create`` and ``from_config``."""

    def __init__(self, package: str) -> None:
        self.package = package
        self._factories: dict[str, Callable[..., Any]] = {}
        self._aliases: dict[str, list[str]] = {}  # canonical name -> aliases
        self._alias_of: dict[str, str] = {}  # alias -> canonical name

    @staticmethod
    def _key(name: str) -> str:
        return name.replace("-", "_")

    # -- registration ---------------------------------------------------------------
    def register(self, name: str, factory: Callable[..., Any] | None = None, aliases: list[str] | None = None):
        """``@registry.register("name")`` or ``registry.register("name", factory, aliases=[...])``.

        A name (or alias) registered twice raises ``AssertionError``
        (relied upon at R: filters/__init__.py:30-33).
        """
        name = self._key(name)

        def _add(fac: Callable[..., Any]) -> Callable[..., Any]:
            assert name not in self._factories and name not in self._alias_of, f"{name} is already registered in {self.package}"
            self._factories[name] = fac
            for alias in aliases or []:
                alias_key = self._key(alias)
                assert alias_key not in self._factories and alias_key not in self._alias_of, (
                    f"{alias_key} is already registered in {self.package}"
                )
                self._alias_of[alias_key] = name
                self._aliases.setdefault(name, []).append(alias_key)
            return fac

        if factory is None:
            return _add
        _add(factory)
        return None

    # -- lookup ---------------------------------------------------------------------
    @property
    def factories(self) -> dict[str, Callable[..., Any]]:
        return dict(self._factories)

    @property
    def registered(self) -> list[str]:
        return sorted(self._factories)

    def aliases(self) -> dict[str, list[str]]:
        return {k: list(v) for k, v in self._aliases.items()}

    def is_registered(self, name: str) -> bool:
        name = self._key(name)
        return name in self._factories or name in self._alias_of

    def lookup(self, name: str, return_none: bool = False) -> Callable[..., Any] | None:
        key = self._key(name)
        key = self._alias_of.get(key, key)
        if key in self._factories:
            return self._factories[key]
        if return_none:
            return None
        raise ValueError(f"Cannot load '{name}' from {self.package}: registered names are {self.registered}")

    # -- construction ---------------------------------------------------------------
    def create(self, name: str, *args: Any, **kwargs: Any) -> Any:
        return self.lookup(name)(*args, **kwargs)

    def from_config(self, config: Any, *args: Any, **kwargs: Any) -> Any:
        """``"name"`` or ``{"name": {kwargs}}`` (YAML recipe entries, R: filters/__init__.py:58)."""
        if isinstance(config, str):
            return self.create(config, *args, **kwargs)
        if not isinstance(config, dict) or len(config) != 1:
            raise ValueError(f"Entry '{config}' must be a name or a dict with a single key, in {self.package}")
        (name, value), = config.items()
        if value is None:
            value = {}
        if isinstance(value, dict):
            return self.create(name, *args, **{**value, **kwargs})
        if isinstance(value, (list, tuple)):
            return self.create(name, *value, *args, **kwargs)
        return self.create(name, value, *args, **kwargs)


filter_registry = Registry("anemoi_transform_amd.filters")
workflow_registry = Registry("anemoi_transform_amd.workflows")
source_registry = Registry("anemoi_transform_amd.sources")


# =================================================================================
# Transform
# =================================================================================
class _TransformMeta(ABCMeta):
    @property
    def reversed(cls) -> Callable[..., "ReversedTransform"]:
        """``Cls.reversed(**cfg)`` builds ``ReversedTransform(Cls(**cfg))`` (R: transform.py:33-44)."""

        def make(*args: Any, **kwargs: Any) -> "ReversedTransform":
            return ReversedTransform(cls(*args, **kwargs))

        make.__name__ = f"{cls.__name__}_reversed"
        make.__doc__ = cls.__doc__
        return make


class Transform(ABC, metaclass=_TransformMeta):
    """Base of everything that maps a FieldList (or DataFrame) to another one."""

    context: Any = None

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}()"

    def __call__(self, data: Any) -> Any:
        return self.forward(data)

    @abstractmethod
    def forward(self, data: Any) -> Any: ...

    def backward(self, data: Any) -> Any:
        raise NotImplementedError(f"{self} is not reversible.")

    def reverse(self) -> "Transform":
        return ReversedTransform(self)

    def __or__(self, other: "Transform") -> "Transform":
        # nested `a | b | c` builds nested two-element pipelines, as in R: transform.py:116-131
        return workflow_registry.create("pipeline", filters=[self, other])

    def patch_data_request(self, data_request: dict) -> dict:
        return data_request

    def reversed(self, *args: Any, **kwargs: Any) -> "Transform":
        return self.__class__.reversed(*args, **kwargs)


class ReversedTransform(Transform):
    """forward <-> backward swapped (R: transform.py:175-244)."""

    def __init__(self, filter: Transform) -> None:
        self.filter = filter

    def __repr__(self) -> str:
        return f"Reversed({self.filter})"

    def forward(self, x: Any) -> Any:
        return self.filter.backward(x)

    def backward(self, x: Any) -> Any:
        return self.filter.forward(x)

    def patch_data_request(self, data_request: dict) -> dict:
        return self.filter.patch_data_request(data_request)


# =================================================================================
# Filters
# =================================================================================
class Filter(Transform):
    """A transform over gridded fields."""


def _is_fieldlist(data: Any) -> bool:
    from .fields import FieldList

    if isinstance(data, FieldList):
        return True
    try:  # real earthkit FieldLists are accepted when earthkit-data is installed
        import earthkit.data as ekd  # type: ignore

        return isinstance(data, ekd.FieldList)
    except ImportError:
        return False


def _is_dataframe(data: Any) -> bool:
    try:
        import pandas as pd
    except ImportError:
        return False
    return isinstance(data, pd.DataFrame)


class DispatchingFilter(Transform):
    """Routes FieldLists to ``*_fields`` and DataFrames to ``*_tabular`` (R: filter.py:35-99).

    A subclass must override ``forward_fields`` or ``forward_tabular``, and may not
    override a ``backward_*`` without its ``forward_*`` (``TypeError`` at class
    creation).  Unknown containers raise ``TypeError`` forward and
    ``NotImplementedError`` backward, as in the reference.
    """

    def __init_subclass__(cls, **kwargs: Any) -> None:
        super().__init_subclass__(**kwargs)

        def overridden(name: str) -> bool:
            return getattr(cls, name) is not getattr(DispatchingFilter, name)

        if not (overridden("forward_fields") or overridden("forward_tabular")):
            raise TypeError(f"{cls.__name__} must override at least one of `forward_fields` or `forward_tabular`")
        for kind in ("fields", "tabular"):
            if overridden(f"backward_{kind}") and not overridden(f"forward_{kind}"):
                raise TypeError(f"{cls.__name__} overrides `backward_{kind}` but not `forward_{kind}`")

    def forward(self, data: Any) -> Any:
        if _is_fieldlist(data):
            return self.forward_fields(data)
        if _is_dataframe(data):
            return self.forward_tabular(data)
        return self.forward_fallback(data)

    def backward(self, data: Any) -> Any:
        if _is_fieldlist(data):
            return self.backward_fields(data)
        if _is_dataframe(data):
            return self.backward_tabular(data)
        return self.backward_fallback(data)

    def forward_fallback(self, data: Any) -> Any:
        raise TypeError(f"No forward method for {type(data)}")

    def backward_fallback(self, data: Any) -> Any:
        raise NotImplementedError(f"No backward method for {type(data)}")

    def forward_fields(self, data: Any) -> Any:
        return self.forward_fallback(data)

    def forward_tabular(self, data: Any) -> Any:
        return self.forward_fallback(data)

    def backward_fields(self, data: Any) -> Any:
        return self.backward_fallback(data)

    def backward_tabular(self, data: Any) -> Any:
        return self.backward_fallback(data)


class SingleFieldFilter(Filter):
    """Transforms fields one at a time; declarative inputs (R: filter.py:102-202).

    ``required_inputs`` / ``optional_inputs`` are validated with the reference's
    messages (pinned by R: tests/test_filter.py:37,50,63); configuration values
    are readable as attributes.  This generic base maps ``forward_transform`` over
    the selected fields, so user-defined subclasses written against the reference
    work unchanged; the built-in per-point filters of this package override
    ``forward`` / ``backward`` with one stack-level kernel launch instead
    (``filters/pointwise.py``).
    """

    required_inputs: tuple[str, ...] | list[str] | None = None
    optional_inputs: dict[str, Any] = {}

    def __init__(self, **kwargs: Any) -> None:
        self._config = {**self.optional_inputs, **kwargs}
        self._validate_inputs()
        self.prepare_filter()
        from .fields import FieldSelection

        self._forward_selection = FieldSelection(**self.forward_select())
        self._backward_selection = FieldSelection(**self.backward_select())

    # -- hooks ------------------------------------------------------------------------
    def prepare_filter(self) -> None:
        pass

    def forward_select(self) -> dict[str, Any]:
        return {}

    def backward_select(self) -> dict[str, Any]:
        return self.forward_select()

    @abstractmethod
    def forward_transform(self, field: Any) -> Any: ...

    def backward_transform(self, field: Any) -> Any:
        raise NotImplementedError("Field backward transform not implemented.")

    def new_field_from_numpy(self, array: Any, *, template: Any, **metadata: Any) -> Any:
        from .fields import new_field_from_numpy

        return new_field_from_numpy(array, template=template, **metadata)

    # -- plumbing ---------------------------------------------------------------------
    def _validate_inputs(self) -> None:
        if not self.required_inputs:
            return
        if not isinstance(self.required_inputs, (list, tuple)):
            raise TypeError("Required inputs must be a list or tuple.")
        if not all(name in self._config for name in self.required_inputs):
            raise TypeError(f"Missing required input(s): '{set(self.required_inputs) - set(self._config)}'.")
        leftover = set(self._config) - (set(self.required_inputs) | set(self.optional_inputs))
        if leftover:
            raise ValueError(f"Unknown input(s): '{leftover}'.")

    @property
    def config(self) -> dict[str, Any]:
        return self._config

    def __getattr__(self, name: str) -> Any:
        # only reached when normal lookup fails: configuration values as attributes
        config = self.__dict__.get("_config")
        if config is not None and name in config:
            return config[name]
        raise AttributeError(f"{type(self).__name__!s} has no attribute or input '{name}'")

    @staticmethod
    def _map_transform(fn: Callable[[Any], Any], fields: Any) -> Any:
        from .fields import new_fieldlist_from_list

        return new_fieldlist_from_list([fn(field) for field in fields])

    def forward(self, data: Any) -> Any:
        sel = self._forward_selection
        return self._map_transform(lambda f: self.forward_transform(f) if sel.match(f) else f, data)

    def backward(self, data: Any) -> Any:
        sel = self._backward_selection
        return self._map_transform(lambda f: self.backward_transform(f) if sel.match(f) else f, data)


# =================================================================================
# Workflows and sources
# =================================================================================
class Workflow(Transform):
    def __iter__(self):
        return iter(self(None))

    def __call__(self, data: Any) -> Any:
        return self.forward(data)


@workflow_registry.register("pipeline")
class Pipeline(Workflow):
    """Filters applied in sequence; backward runs them in reverse (R: workflows/pipeline.py:18-64)."""

    def __init__(self, *, filters: list[Any]) -> None:
        self.filters = filters

    def __repr__(self) -> str:
        return "Pipeline(" + " | ".join(repr(f) for f in self.filters) + ")"

    def forward(self, data: Any) -> Any:
        # same result as `for f in filters: data = f.forward(data)`; runs of per-point filters
        # (optionally behind a regrid) are collapsed into one kernel launch per stack
        from .filters.fusion import forward_fused

        return forward_fused(self.filters, data)

    def backward(self, data: Any) -> Any:
        # same result as `for f in reversed(filters): data = f.backward(data)` (R: workflows/pipeline.py:50-64), fused the
        # same way: the backward of a filter is the forward of its ReversedTransform
        from .filters.fusion import flatten, forward_fused

        return forward_fused([ReversedTransform(f) for f in reversed(flatten(self.filters))], data)


class Source(Transform):
    """Provides data; iterating a source iterates its fields (R: source.py)."""

    def __iter__(self):
        return iter(self.forward())


def create_source(context: Any, config: Any) -> Any:
    source = source_registry.from_config(config)
    source.context = context
    return source
