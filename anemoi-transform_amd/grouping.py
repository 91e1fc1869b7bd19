"""Grouping of fields that belong together (same date / level / member, different ``param``).

Mirror of R: grouping/__init__.py:55-175 for the multi-input filters: the grouping key of a
field is its MARS namespace minus ``param`` (and ``variable``); fields without a MARS
namespace fall back to all their metadata keys except coordinates and values
(R: grouping/__init__.py:70-91).  Pure host bookkeeping — the arithmetic of a group is done
for ALL groups at once by ``atx_combine_stack`` (filters/multi.py).
"""

from __future__ import annotations

from collections import defaultdict
from typing import Any, Callable, Iterator


def _lost(f: Any) -> None:
    raise ValueError(f"Lost field {f}")


def _flatten(params: Any) -> list[str]:
    flat: list[str] = []
    for p in params:
        if isinstance(p, (list, tuple)):
            flat.extend(_flatten(p))
        else:
            flat.append(p)
    return flat


def grouping_key(field: Any, extract: list[str], remove: list[str] | None = None) -> tuple[dict, dict]:
    """``(key, extracted)``: the identifying metadata of a field and the values pulled out of it."""
    key = dict(field.metadata(namespace="mars") or {})
    if not key:
        names = [k for k in field.metadata().keys() if k not in ("latitudes", "longitudes", "values")]
        key = {k: field.metadata(k) for k in names}
        if not names:
            raise NotImplementedError(f"GroupByParam: {field} has no sufficient metadata")
    extracted = {}
    for name in extract:
        extracted[name] = key.pop(name, field.metadata().get(name, None))
    for name in remove or []:
        key.pop(name, None)
    return key, extracted


class GroupByParam:
    """Yield tuples of fields, one per requested ``param``, that share every other key."""

    def __init__(self, params: Any) -> None:
        if not isinstance(params, (list, tuple)):
            params = [params]
        self.params = _flatten(params)

    def _get_groups(self, data: Any, *, other: Callable[[Any], None] = _lost) -> None:
        assert callable(other), type(other)
        self.groups: dict[frozenset, dict[str, Any]] = defaultdict(dict)
        self.groups_params: set[str] = set()
        for f in data:
            key, extras = grouping_key(f, ["param"], ["variable"])
            param = extras["param"]
            if param not in self.params:
                other(f)
                continue
            frozen = frozenset(key.items())
            if param in self.groups[frozen]:
                raise ValueError(f"Duplicate component {param} for {frozen}")
            self.groups[frozen][param] = f
            self.groups_params.add(param)

    def iterate(self, data: Any, *, other: Callable[[Any], None] = _lost) -> Iterator[tuple[Any, ...]]:
        self._get_groups(data, other=other)
        for group in self.groups.values():
            if len(group) != len(self.params):
                raise ValueError(f"Missing component. Want {sorted(self.params)}, got {sorted(group.keys())}")
            yield tuple(group[p] for p in self.params)


class GroupByParamVertical(GroupByParam):
    """As ``GroupByParam`` but all levels of a parameter are collected into one FieldList
    (R: grouping/__init__.py:140-175)."""

    def _get_groups(self, data: Any, *, other: Callable[[Any], None] = _lost) -> None:
        from .fields import FieldList

        assert callable(other), type(other)
        self.groups = defaultdict(dict)
        self.groups_params = set()
        levels: dict[str, list] = defaultdict(list)
        for f in data:
            key, extras = grouping_key(f, ["param", "levelist"], ["variable", "levtype"])
            param, level = extras["param"], extras["levelist"]
            if param not in self.params:
                other(f)
                continue
            frozen = frozenset(key.items())
            if level is None:
                if param in self.groups[frozen]:
                    raise ValueError(f"Duplicate component {param} for {frozen}")
                self.groups[frozen][param] = f
            else:
                if param in self.groups[frozen]:
                    if level in levels[param]:
                        raise ValueError(f"Duplicate component {param} for {frozen} and level {level}")
                    self.groups[frozen][param].append(f)
                else:
                    self.groups[frozen][param] = FieldList([f])
                levels[param].append(level)
            self.groups_params.add(param)
