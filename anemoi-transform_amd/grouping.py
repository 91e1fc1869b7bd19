"""Grouping of fields that belong together (same date / level / member, different ``param``).

Host bookkeeping for the multi-input filters, with the contract of R: grouping/__init__.py:55-175: the identity of a
field is its MARS namespace minus ``param`` (and ``variable``); fields without a MARS namespace are identified by all
their metadata except coordinates and values (R: grouping/__init__.py:70-91); one group = one field per requested
parameter sharing that identity, yielded in first-seen order; a field of another parameter goes to ``other``; duplicates
and incomplete groups are errors with the reference's messages.  The arithmetic of a group is done for ALL groups at once
by ``atx_combine_stack`` (filters/multi.py).
"""

from __future__ import annotations

from typing import Any, Callable, Iterator

_NOT_IDENTITY = ("latitudes", "longitudes", "values")


def _lost(f: Any) -> None:
    raise ValueError(f"Lost field {f}")


def _names(params: Any) -> list[str]:
    """Parameter names of an arbitrarily nested list / tuple, depth first."""
    if not isinstance(params, (list, tuple)):
        return [params]
    return [name for p in params for name in _names(p)]


def identity_of(field: Any, take: tuple[str, ...], drop: tuple[str, ...] = ()) -> tuple[frozenset, dict[str, Any]]:
    """``(identity, taken)``: the hashable identity of a field without the keys in ``take`` / ``drop``, and the values of
    the ``take`` keys (looked up in the field's full metadata when the MARS namespace does not carry them)."""
    described = dict(field.metadata(namespace="mars") or {})
    if not described:
        keys = [k for k in field.metadata().keys() if k not in _NOT_IDENTITY]
        if not keys:
            raise NotImplementedError(f"GroupByParam: {field} has no sufficient metadata")
        described = {k: field.metadata(k) for k in keys}
    taken = {k: described.pop(k) if k in described else field.metadata().get(k, None) for k in take}
    for k in drop:
        described.pop(k, None)
    return frozenset(described.items()), taken


class GroupByParam:
    """Tuples of fields, one per requested ``param`` (in the order requested), that share every other key."""

    TAKE: tuple[str, ...] = ("param",)
    DROP: tuple[str, ...] = ("variable",)

    def __init__(self, params: Any) -> None:
        self.params = _names(params)

    # -- one field joins the group of its identity -------------------------------------------------------
    def _place(self, members: dict[str, Any], identity: frozenset, field: Any, taken: dict[str, Any]) -> None:
        param = taken["param"]
        if param in members:
            raise ValueError(f"Duplicate component {param} for {identity}")
        members[param] = field

    def _collect(self, data: Any, other: Callable[[Any], None]) -> dict[frozenset, dict[str, Any]]:
        assert callable(other), type(other)
        groups: dict[frozenset, dict[str, Any]] = {}
        self._seen_levels: dict[str, list] = {}
        for field in data:
            identity, taken = identity_of(field, self.TAKE, self.DROP)
            if taken["param"] not in self.params:
                other(field)
            else:
                self._place(groups.setdefault(identity, {}), identity, field, taken)
        self.groups = groups
        self.groups_params = {p for members in groups.values() for p in members}
        return groups

    def iterate(self, data: Any, *, other: Callable[[Any], None] = _lost) -> Iterator[tuple[Any, ...]]:
        for members in self._collect(data, other).values():
            if len(members) != len(self.params):
                raise ValueError(f"Missing component. Want {sorted(self.params)}, got {sorted(members.keys())}")
            yield tuple(members[p] for p in self.params)


class GroupByParamVertical(GroupByParam):
    """As ``GroupByParam`` with the level left out of the identity: all levels of a parameter are collected, in input
    order, into one FieldList per group (R: grouping/__init__.py:140-175); a field without a level stays a single field."""

    TAKE = ("param", "levelist")
    DROP = ("variable", "levtype")

    def _place(self, members: dict[str, Any], identity: frozenset, field: Any, taken: dict[str, Any]) -> None:
        from .fields import FieldList

        param, level = taken["param"], taken["levelist"]
        if level is None:
            return super()._place(members, identity, field, taken)
        seen = self._seen_levels.setdefault(param, [])
        if param not in members:
            members[param] = FieldList([field])
        elif level in seen:
            raise ValueError(f"Duplicate component {param} for {identity} and level {level}")
        else:
            members[param].append(field)
        seen.append(level)
