"""Filters whose arithmetic is NOT pinned by the reference say so when they are constructed (one warning per filter instance), the way
`regrid`'s default route does (filters/regrid.py): the wind and humidity conversions restate earthkit-meteo (absent here, SURVEY.md
§8c (4)) and `convert` knows one pinned unit pair, K <-> degC (R: tests/field_filters/test_rescale.py:58-72).  The filters that ARE
pinned stay silent.  README.md's parity table is generated from DESIGN.md §7 and lists the same split."""

from __future__ import annotations

import logging
import os
import re

import pytest

from anemoi_transform_amd.filters import create_filter_by_name

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

UNPINNED = ["uv_to_ddff", "ddff_to_uv", "r_to_d", "d_to_r", "q_to_r", "r_to_q", "q_to_r_height_with_p", "r_to_q_height_with_p"]


def parity_warnings(caplog, name, fresh=True, words=("pinned",), **config):
    """The WARNING-level parity notes of one construction; `fresh`: as the first construction of the process (core.say_once)."""
    from anemoi_transform_amd.core import reset_notes

    if fresh:
        reset_notes()
    caplog.clear()
    with caplog.at_level(logging.DEBUG, logger="anemoi_transform_amd"):
        create_filter_by_name(name, **config)
    return [r.getMessage() for r in caplog.records if r.levelno >= logging.WARNING and any(w in r.getMessage() for w in words)]


@pytest.mark.parametrize("name", UNPINNED)
def test_filters_on_restated_earthkit_meteo_arithmetic_say_so(caplog, name):
    notes = parity_warnings(caplog, name)
    assert len(notes) == 1, notes  # said by the first instance of the class in a process ...
    assert "earthkit-meteo" in notes[0] and "reference's test points" in notes[0] and "np.allclose" in notes[0]
    assert parity_warnings(caplog, name, fresh=False) == []  # ... and not again by the next (a pipeline that builds its filters per date)
    assert any("earthkit-meteo" in r.getMessage() and r.levelno == logging.DEBUG for r in caplog.records)  # still there for who asks


def test_convert_says_which_pairs_are_not_pinned(caplog):
    assert parity_warnings(caplog, "convert", unit_in="K", unit_out="degC", param="2t") == []  # the pinned pair, either way round
    assert parity_warnings(caplog, "convert", unit_in="degC", unit_out="K", param="2t") == []
    notes = parity_warnings(caplog, "convert", unit_in="Pa", unit_out="hPa", param="sp")
    assert len(notes) == 1 and "Pa -> hPa" in notes[0] and "K <-> degC" in notes[0]
    assert ("private unit table" in notes[0]) or ("pint" in notes[0])
    assert parity_warnings(caplog, "convert", fresh=False, unit_in="Pa", unit_out="hPa", param="sp") == []  # once per unit pair and process
    assert len(parity_warnings(caplog, "convert", fresh=False, unit_in="hPa", unit_out="Pa", param="sp")) == 1  # another pair: its own note


def test_convert_with_pint_installed_informs_instead_of_warning(caplog, monkeypatch):
    """pint is the reference's own converter (R: rescale.py:94): when it supplies the factor the note is INFO, not WARNING."""
    import sys
    import types

    class Quantity:
        def __init__(self, x, unit):
            self.x, self.unit = x, unit

        def to(self, unit):
            assert (self.unit, unit) == ("Pa", "hPa")
            return types.SimpleNamespace(magnitude=self.x / 100.0)

    fake = types.ModuleType("pint")
    fake.UnitRegistry = lambda: types.SimpleNamespace(Quantity=Quantity)
    monkeypatch.setitem(sys.modules, "pint", fake)
    assert parity_warnings(caplog, "convert", unit_in="Pa", unit_out="hPa", param="sp") == []
    said = [r for r in caplog.records if "K <-> degC" in r.getMessage()]
    assert len(said) == 1 and said[0].levelno == logging.INFO and "the reference's own converter" in said[0].getMessage()


def test_snow_cover_says_where_it_can_differ(caplog):
    """`snow_cover` is pinned — and has a jump in its statement (R: snow_cover.py:38, `> 0.99 -> 1`) that makes the last bit of tanh
    visible: the filter says so (once per process), README's parity table says so, tests/test_gpu_random_shapes.py holds it to that."""
    notes = parity_warnings(caplog, "snow_cover", words=("0.99",))
    assert len(notes) == 1 and "snow_cover.py:38" in notes[0] and "either side is the statement's own value" in notes[0]
    assert parity_warnings(caplog, "snow_cover", fresh=False, words=("0.99",)) == []
    readme = open(os.path.join(ROOT, "README.md")).read()
    row = next(ln for ln in readme.splitlines() if ln.startswith("|") and "snow_cover" in ln)
    assert "except at the jump" in row and "either side is the statement's own value" in row


@pytest.mark.parametrize("name,config", [
    ("rescale", dict(scale=1.0, offset=-273.15, param="2t")), ("orog_to_z", {}), ("lnsp_to_sp", {}), ("snow_depth_m", {}), ("cos_sin_from_rad", dict(param="mwd")),
    ("rodeo_opera_clipping", {}), ("sum", dict(params=["a", "b"], output="c")), ("regrid", dict(in_grid="O32", out_grid=[5.0, 5.0], method="nearest")),
])
def test_pinned_filters_are_silent(caplog, name, config):
    assert parity_warnings(caplog, name, **config) == []


def test_readme_parity_table_is_generated_from_design():
    """README.md's table repeats DESIGN.md §7 row for row (tools/design_tables.py, block `filter-parity`), and every filter that warns
    at run time is marked there."""
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    readme = open(os.path.join(ROOT, "README.md")).read()
    block = re.search(r"<!-- BEGIN generated: filter-parity[^\n]*-->\n(.*?)\n<!-- END generated: filter-parity -->", readme, re.S)
    assert block, "README.md has no generated filter-parity block"
    table = [ln for ln in block.group(1).splitlines() if ln.startswith("|")]
    section = design[design.index("| Reference file | Here | Parity | Says so at run time |"):]
    rows = [ln for ln in section[:section.index("\n\n")].splitlines()[2:]]
    assert len(table) == len(rows) + 2 and len(rows) >= 10  # header + rule + the rows
    for ln in rows:
        assert ln in block.group(1), ln
    warned = "\n".join(ln for ln in table if "| yes" in ln)
    for name in ("uv_to_ddff", "dewpoint", "q_to_r", "q_height", "convert"):
        assert name in warned, name
