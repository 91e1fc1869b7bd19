"""BASELINE.json configs[4] and configs[3] at FULL SIZE through the drop-in API — the route a user of the reference takes:
``create_filter_by_name(...)``, ``a | b | c`` and ``Filter.forward(fieldlist)`` (R: filters/__init__.py:36-60,
workflows/pipeline.py:46-48, filters/fields/regrid.py:174-208, orog_to_z.py:57-60, rescale.py:101-105).

tests/test_gpu_configs.py reaches the same kernels through ``GatherPlan.apply_many`` and hand-built ``native.level_program``
tables; tests/test_fusion.py drives the plugin route on O16 / O32 grids with a handful of fields.  Here the code between the two —
``filters/fusion.py`` (stage folding, one program per stack), ``fields.group_into_stacks`` (device stacks used in place, host lists
regrouped by variable and cut at ``MAX_STACK_LEVELS``), the runs-of-levels / per-level-table choice of ``native.level_program`` and
``RegridFilter(shard=...)`` — runs on 137 and 3 x 137 float64 O2560 fields and on the 3 288-field float32 O1280 list, against the
oracle's statements.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

from anemoi_transform_amd import interp, native
from anemoi_transform_amd.fields import ArrayField, FieldList, fieldlist_from_dicts, new_field_from_stack
from anemoi_transform_amd.filters import create_filter_by_name
from anemoi_transform_amd.gather import GatherPlan
from anemoi_transform_amd.grids import lookup
from anemoi_transform_amd.stack import COLUMNS, Stack
from oracle import oracle

pytestmark = pytest.mark.gpu

N_LEV = 137
GATHERS = ("regrid_ell", "regrid_csr", "regrid_ell_batch")


@pytest.fixture
def launches(monkeypatch):
    """Counts kernel launches by wrapping the native entry points (as tests/test_fusion.py does)."""
    counts = {name: 0 for name in GATHERS + ("pointwise_stack", "select_levels", "relayout")}
    depth = [0]  # only the OUTERMOST wrapper call is a launch: `native.regrid_ell(tgt_rows=...)` hands an ordered plan to `regrid_ell_batch`
    for name in counts:
        real = getattr(native, name)

        def wrapped(*a, _real=real, _name=name, **k):
            if depth[0] == 0:
                counts[_name] += 1
            depth[0] += 1
            try:
                return _real(*a, **k)
            finally:
                depth[0] -= 1

        monkeypatch.setattr(native, name, wrapped)
    return counts


def gathers(counts) -> int:
    return sum(counts[name] for name in GATHERS)


def bits(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous().view(torch.int64 if t.dtype == torch.float64 else torch.int32)


def field_bits(f) -> torch.Tensor:
    stack, level = f.stack_ref()
    return bits(stack.level_view(level))


# ------------------------------------------------------------------------------------------------------------------------------
# configs[4]: regrid | orog_to_z | convert on ERA5-shape O2560 fields, float64 (the reference's own width, R: fields.py:178-202)
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def o2560(dev, tmp_path_factory):
    src, tgt = lookup("o2560"), lookup("0.25")
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    assert (n_src, n_tgt) == (26306560, 1038240)
    idx, w = interp.knn_inverse_distance(src, tgt, k=4, device=True)  # cKDTree's table, built on the device (tests/test_gpu_knn.py)
    path = str(tmp_path_factory.mktemp("matrices") / "o2560-to-0p25-knn4.npz")
    interp.save_matrix_npz(path, interp.ell_to_csr(idx, w, n_src), src, tgt)  # the reference's own file layout (R: regrid.py:281-290)
    return dict(src=src, tgt=tgt, n_src=n_src, n_tgt=n_tgt, idx=idx, w=w, path=path, indptr=np.arange(n_tgt + 1) * 4)


def config5_pipeline(path):
    """BASELINE configs[4] as a user writes it (R: workflows/pipeline.py:33-48: `a | b | c` nests two-element pipelines)."""
    return (create_filter_by_name("regrid", matrix=path) | create_filter_by_name("orog_to_z")
            | create_filter_by_name("convert", unit_in="K", unit_out="degC", param="t"))


def device_fields(stack, names, src):
    """One device-resident field per level of ``stack``; ``names[l]`` is its param, levelist counts within the param."""
    template = ArrayField(np.zeros(1), {"param": "?", "units": "K"}, np.zeros(1), np.zeros(1))
    seen: dict[str, int] = {}
    fields = []
    for level, name in enumerate(names):
        seen[name] = seen.get(name, 0) + 1
        fields.append(new_field_from_stack(stack, level, template=template, latitudes=src["latitudes"], longitudes=src["longitudes"],
                                           metadata={"param": name, "levelist": seen[name]}))
    return FieldList(fields)


def check_config5(case, x, names, launches, monkeypatch, samples):
    """Fused pipeline over the fields of ``x``: one gather launch, the filter-by-filter bits, the oracle's bits on ``samples``."""
    fields = device_fields(x, names, case["src"])
    before = dict(launches)
    fused = config5_pipeline(case["path"]).forward(fields)
    made = {k: launches[k] - before[k] for k in launches}
    assert sum(made[g] for g in GATHERS) == 1 and made["pointwise_stack"] == 0, made  # ONE launch, values stored once
    assert made["select_levels"] == 0 and made["relayout"] == 0, made  # the resident stack is used in place: no copy of 29 / 87 GB
    assert len(fused) == len(names)

    monkeypatch.setenv("ATX_NO_FUSION", "1")
    unfused = config5_pipeline(case["path"]).forward(fields)  # R: pipeline.py:46-48 — one full FieldList between every two filters
    monkeypatch.delenv("ATX_NO_FUSION")
    assert launches["pointwise_stack"] - before["pointwise_stack"] >= 2

    expect_param = ["z" if n == "orog" else n for n in names]
    assert [f.metadata("param") for f in fused] == expect_param == [f.metadata("param") for f in unfused]
    assert [f.metadata("levelist") for f in fused] == [f.metadata("levelist") for f in fields]  # list order kept
    for a, b, name in zip(fused, unfused, names):
        stack, _ = a.stack_ref()  # still in HBM, float64, on the target grid
        assert stack.data.dtype == torch.float64 and stack.n_pts == case["n_tgt"] and a.shape == (case["n_tgt"],)
        assert torch.equal(field_bits(a), field_bits(b)), (name, a.metadata("levelist"))
        assert a.metadata("units") == b.metadata("units") == ("degC" if name == "t" else a.metadata("units"))
    assert fused[names.index("t")].metadata("units") == "degC"
    lat, lon = fused[0].grid_points()
    assert np.array_equal(lat, case["tgt"]["latitudes"]) and np.array_equal(lon, case["tgt"]["longitudes"])
    lat, lon = fused[-1].grid_points()
    assert np.array_equal(lat, case["tgt"]["latitudes"]) and np.array_equal(lon, case["tgt"]["longitudes"])

    weights = case["w"].reshape(-1)
    for pos in samples:
        base = oracle.csr_apply(weights, case["idx"].reshape(-1), case["indptr"], (case["n_tgt"], case["n_src"]), x.level_numpy(pos))
        if names[pos] == "t":
            want = oracle.rescale_forward(base, 1.0, -273.15)  # R: rescale.py:25, scale / offset of K -> degC (R: rescale.py:101-105)
        elif names[pos] == "orog":
            want = oracle.orog_to_z(base)  # R: orog_to_z.py:59
        else:
            want = base  # a variable no filter selects leaves the pipeline as the regrid made it
        got = fused[pos].to_numpy()
        assert got.dtype == np.float64 and want.dtype == np.float64
        assert np.array_equal(got, want), (pos, names[pos])  # the oracle's bits through Pipeline.forward()


def test_config5_pipeline_137_float64_fields_through_the_plugin_api(o2560, dev, launches, monkeypatch):
    """137 device-resident float64 O2560 fields — 136 x `t` and one `orog` in their midst — through
    `regrid(matrix=<k = 4 npz>) | orog_to_z | convert(K -> degC, param=t)`: ONE gather launch with the two per-point filters in its
    epilogue, every output field bit-equal to the filter-by-filter pipeline, three fields equal to the oracle's
    `csr_array @ x` then `x * 1 + -273.15` / `x * g`."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(2560)
    x = Stack.empty(o2560["n_src"], N_LEV, torch.float64, dev, COLUMNS, zero=True)
    x.data[:, :N_LEV].normal_(270.0, 15.0, generator=gen)
    orog_at = 40
    names = ["t"] * orog_at + ["orog"] + ["t"] * (N_LEV - orog_at - 1)
    x.data[:, orog_at].uniform_(-50.0, 5500.0, generator=gen)
    check_config5(o2560, x, names, launches, monkeypatch, samples=(0, orog_at, N_LEV - 1))
    del x
    torch.cuda.empty_cache()


def test_config5_pipeline_three_variables_float64_through_the_plugin_api(o2560, dev, launches, monkeypatch):
    """SURVEY.md §8d's wording of configs[4] — 137 levels x {t, orography-like, one more variable} — as ONE resident 411-level float64
    stack (86.5 GB): `t` is converted, `orog` becomes `z`, `q` is selected by no filter and must leave with the regrid's own bits."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(2561)
    n_lev = 3 * N_LEV
    x = Stack.empty(o2560["n_src"], n_lev, torch.float64, dev, COLUMNS, zero=True)
    x.data[:, :n_lev].normal_(270.0, 15.0, generator=gen)
    names = ["t"] * N_LEV + ["orog"] * N_LEV + ["q"] * N_LEV
    check_config5(o2560, x, names, launches, monkeypatch, samples=(5, N_LEV + 70, n_lev - 1))
    del x
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------------------------------------
# configs[3]: O1280 -> N320-sized, 6 variables x 137 levels x 4 timesteps, target points sharded 8 ways
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def o1280_n320(dev, tmp_path_factory):
    src, tgt = lookup("o1280"), lookup("n320-sized")
    n_src, n_tgt = len(src["latitudes"]), len(tgt["latitudes"])
    assert (n_src, n_tgt) == (6599680, 542080)
    idx, w = interp.knn_inverse_distance(src, tgt, k=4)  # cKDTree: the reference's own builder
    path = str(tmp_path_factory.mktemp("matrices") / "o1280-to-n320-knn4.npz")
    interp.save_matrix_npz(path, interp.ell_to_csr(idx, w, n_src), src, tgt)
    return dict(src=src, tgt=tgt, n_src=n_src, n_tgt=n_tgt, idx=idx, w=w, path=path, indptr=np.arange(n_tgt + 1) * 4)


def test_config4_3288_fields_through_a_sharded_regrid_filter(o1280_n320, dev, launches):
    """The 3 288-field list (24 resident float32 stacks of 137 levels, 88.7 GB) through
    `create_filter_by_name("regrid", matrix=..., shard=(r, 8))` for ranks 0, 3 and 7 of the 8-way target partition: the list comes
    level by level with the variables and timesteps interleaved (as a GRIB stream does), every stack is used in place, the 24 stacks go
    through the batched launch, each output field holds the bits of `plan.shard(r, 8).apply_many`, five fields are compared with scipy on
    the rank's window, and the fields know which target points they hold."""
    case = o1280_n320
    n_stack, params = 24, ["t", "q", "u", "v", "w", "z"]
    gen = torch.Generator(device=dev)
    stacks = []
    for sid in range(n_stack):  # stack sid: variable sid // 4 at timestep sid % 4
        gen.manual_seed(4000 + sid)
        st = Stack.empty(case["n_src"], N_LEV, torch.float32, dev, COLUMNS, zero=True)
        st.data[:, :N_LEV].normal_(250.0 + 5.0 * (sid // 4), 20.0, generator=gen)
        stacks.append(st)
    template = ArrayField(np.zeros(1), {"param": "?"}, np.zeros(1), np.zeros(1))
    fields, origin = [], []
    for step in range(4):
        for level in range(N_LEV):
            for v, name in enumerate(params):
                sid = 4 * v + step
                fields.append(new_field_from_stack(stacks[sid], level, template=template, latitudes=case["src"]["latitudes"],
                                                   longitudes=case["src"]["longitudes"],
                                                   metadata={"param": name, "levelist": level + 1, "step": 6 * step}))
                origin.append((sid, level))
    fields = FieldList(fields)
    assert len(fields) == 3288

    plan = GatherPlan(case["n_src"], case["n_tgt"], index=case["idx"], weights=case["w"])
    bounds = plan.bounds(8)
    w32 = case["w"].astype(np.float32).reshape(-1)
    samples = (0, 821, 1644, 2500, 3287)
    wants = {pos: oracle.csr_apply(w32, case["idx"].reshape(-1), case["indptr"], (case["n_tgt"], case["n_src"]),
                                   stacks[origin[pos][0]].level_numpy(origin[pos][1])) for pos in samples}
    for rank in (0, 3, 7):
        before = dict(launches)
        out = create_filter_by_name("regrid", matrix=case["path"], shard=(rank, 8)).forward(fields)
        made = {k: launches[k] - before[k] for k in launches}
        assert gathers(made) == 1 and made["regrid_ell_batch"] == 1, made  # ONE batched call (16 + 8 stacks inside the library)
        assert made["select_levels"] == 0 and made["relayout"] == 0, made  # every resident stack used in place
        lo, hi = bounds[rank], bounds[rank + 1]
        direct = plan.shard(rank, 8).apply_many(stacks)
        assert len(out) == 3288 and direct[0].n_pts == hi - lo
        for pos, f in enumerate(out):
            sid, level = origin[pos]
            assert f.target_range() == (lo, hi, case["n_tgt"])
            stack, at = f.stack_ref()
            assert stack.data.dtype == torch.float32 and stack.n_pts == hi - lo
            assert torch.equal(bits(stack.level_view(at)), bits(direct[sid].level_view(level))), (rank, pos)
        assert [f.metadata("param") for f in out[:12]] == params * 2
        assert out[3287].metadata("step") == 18 and out[3287].metadata("levelist") == 137 and out[6].metadata("levelist") == 2
        lat, lon = out[1234].grid_points()
        assert np.array_equal(lat, case["tgt"]["latitudes"][lo:hi]) and np.array_equal(lon, case["tgt"]["longitudes"][lo:hi])
        for pos in samples:
            got = out[pos].to_numpy()
            assert got.dtype == np.float32 and np.array_equal(got, wants[pos][lo:hi]), (rank, pos)  # scipy's bits on this rank's window
        del out, direct
    del stacks, fields
    torch.cuda.empty_cache()


def test_config4_host_list_is_regrouped_by_variable_and_cut_into_stacks(o1280_n320, dev, launches, monkeypatch, tmp_path):
    """The same filter chain on fields that arrive from the HOST, level by level with four variables interleaved — 548 float32 O1280
    fields, more than one stack holds (`fields.MAX_STACK_LEVELS` = 512): the list is uploaded as "all levels of t, then q, ..."
    (`fields._variables_together`), cut into two stacks that share one batched launch... and, with `convert` on `t` and `clip` on `q`
    and `apply_mask` with a FULL-grid mask file on `u` and `t` behind a sharded regrid, into one fused launch per stack (four stages, a point mask windowed to the rank's
    slice).  Every field against the filter-by-filter run, samples against the oracle."""
    from anemoi_transform_amd import fields as fields_mod

    case = o1280_n320
    params = ["t", "q", "u", "orog"]
    rng = np.random.default_rng(548)
    noise = [rng.standard_normal(case["n_src"]).astype(np.float32) for _ in range(8)]
    base = (270.0 + 25.0 * np.sin(np.deg2rad(case["src"]["latitudes"])) * np.cos(2.0 * np.deg2rad(case["src"]["longitudes"]))).astype(np.float32)
    specs = []
    for level in range(N_LEV):
        for v, name in enumerate(params):
            values = base + np.float32(0.25 * level + 3.0 * v) + noise[(level + 3 * v) % 8] * np.float32(1.0 + 0.01 * v)
            specs.append({"param": name, "levelist": level + 1, "values": values, "latitudes": case["src"]["latitudes"],
                          "longitudes": case["src"]["longitudes"]})
    fields = fieldlist_from_dicts(specs)
    assert len(fields) == 548 > fields_mod.MAX_STACK_LEVELS

    # a FULL-grid point mask behind a sharded regrid: the fields know their window of the target grid and the mask is cut to it
    mask_full = np.random.default_rng(549).random(case["n_tgt"]) < 0.25
    mask_path = str(tmp_path / "target-mask.npy")
    np.save(mask_path, mask_full.astype(np.float64))

    def pipeline():
        return (create_filter_by_name("regrid", matrix=case["path"], shard=(3, 8)) | create_filter_by_name("orog_to_z")
                | create_filter_by_name("convert", unit_in="K", unit_out="degC", param="t")
                | create_filter_by_name("clip", param="q", minimum=280.0, maximum=300.0)
                | create_filter_by_name("apply_mask", path=mask_path, mask_value=1, param=["u", "t"]))

    before = dict(launches)
    fused = pipeline().forward(fields)
    made = {k: launches[k] - before[k] for k in launches}
    assert gathers(made) == 2 and made["pointwise_stack"] == 0, made  # 512 + 36 fields: one fused launch per stack
    monkeypatch.setenv("ATX_NO_FUSION", "1")
    unfused = pipeline().forward(fields)
    monkeypatch.delenv("ATX_NO_FUSION")

    plan = GatherPlan(case["n_src"], case["n_tgt"], index=case["idx"], weights=case["w"])
    lo, hi = plan.bounds(8)[3], plan.bounds(8)[4]
    assert [f.metadata("param") for f in fused] == ["z" if s["param"] == "orog" else s["param"] for s in specs]
    assert [f.metadata("levelist") for f in fused] == [s["levelist"] for s in specs]
    for pos, (a, b) in enumerate(zip(fused, unfused)):
        assert a.target_range() == (lo, hi, case["n_tgt"]) == b.target_range()
        assert torch.equal(field_bits(a), field_bits(b)), pos
    w32 = case["w"].astype(np.float32).reshape(-1)
    for pos in (0, 1, 2, 3, 273, 546, 547):
        name = specs[pos]["param"]
        full = oracle.csr_apply(w32, case["idx"].reshape(-1), case["indptr"], (case["n_tgt"], case["n_src"]), specs[pos]["values"])[lo:hi]
        if name == "t":
            want = oracle.rescale_forward(full, np.float32(1.0), np.float32(-273.15))
        elif name == "orog":
            want = full * np.float32(oracle.G)
        elif name == "q":
            want = oracle.clip(full, np.float32(280.0), np.float32(300.0))
        else:
            want = full
        if name in ("u", "t"):  # R: apply_mask.py:185 `values[self.mask] = np.nan`, on this rank's window of the mask
            want = oracle.apply_mask_values(want.copy(), mask_full[lo:hi])
            assert np.isnan(want).sum() == mask_full[lo:hi].sum() > 0
        got = fused[pos].to_numpy()
        assert got.dtype == np.float32 and np.array_equal(got, want, equal_nan=True), (pos, name)
