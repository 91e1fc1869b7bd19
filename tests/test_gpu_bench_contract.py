"""bench.py's contract (one JSON line with the driver's keys plus `roofline` / `cpu_baseline`) on small grids, and the
N > 1 code path rehearsed with two ranks sharing the one GPU of the test box over gloo (the real N > 1 runs use RCCL,
one GPU per rank; this keeps sharding, exchange, batched launch and max-over-ranks timing from rotting)."""

from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--src-grid", "o96", "--tgt-grid", "1.0", "--levels", "7", "--steps", "3", "--warmup", "1"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline"}


def last_json(stdout: str) -> dict:
    lines = [l for l in stdout.strip().splitlines() if l.strip()]
    # stdout carries exactly ONE line, the JSON one: nothing else (the collective libraries' connection notes go to stderr)
    assert len(lines) == 1 and lines[0].startswith("{"), stdout
    return json.loads(lines[0])


def test_single_gpu_line():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--cpu-seconds", "0.5", "--no-extras"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-2000:]
    d = last_json(run.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    # every line of the N = 1, 2, 4, 8 series does the SAME total work per step (one stack): the series is strong scaling
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["config"]["stacks_per_step"] == 1 and "fixed-total-work" in d["config"]["sharding"]
    assert d["unit"] == "grid-points/s" and d["value"] > 0 and d["dtype"] == "f64" and "workload" in d["config"]  # the reference's own arithmetic (R: fields.py:178-202)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["min_launch_ms"] <= r["median_launch_ms"] and r["min_launch_ms"] <= r["avg_launch_ms"]  # HIP events around single launches
    j = r["job"]  # the whole job on the wall clock of the timed region: at N = 1 the same bytes as the launch
    assert j["algorithmic_bytes_per_step_all_ranks"] == r["algorithmic_bytes_per_launch"] and j["peak"] == 8000.0 and 0 < j["frac"] <= 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    # both reference statements (csr_array @ x, x[..., idx]) in both widths
    assert set(c["variants"]) == {"csr_f64", "csr_f32", "k1_f64", "k1_f32"} and all(v["value"] > 0 for v in c["variants"].values())
    assert c["value"] == c["variants"]["csr_f64"]["value"]
    # what the CPU figures were measured on, inside `cpu_baseline` so that the driver's record keeps it
    host = c["host"]
    assert {"model", "nproc", "cpu_quota_cores", "numpy", "scipy", "python"} <= set(host)
    assert isinstance(host["model"], str) and host["model"] and host["nproc"] >= 1
    import numpy
    import scipy

    assert host["numpy"] == numpy.__version__ and host["scipy"] == scipy.__version__
    assert "multi_gpu" not in d["config"]  # N = 1: nothing to mirror
    assert d["parity_max_rel_err"] == 0.0  # float64: scipy's bits


def test_single_gpu_line_f32():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--dtype", "f32", "--cpu-seconds", "0.5", "--headline-shape-only"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-2000:]
    d = last_json(run.stdout)
    assert d["dtype"] == "f32" and d["parity_max_rel_err"] <= 1e-6
    e = d["extras"]
    assert e["f64_columns"]["value"] > 0 and e["nearest_k1"]["value"] > 0 and e["nearest_k1_f64"]["value"] > 0
    assert set(e["gpu_over_cpu_one_core"]) == {"k4_f64", "k4_f32", "k1_f64", "k1_f32"}


def test_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment — the shape of the driver's N = 1 command — starts its own
    two workers (fresh processes; the parent never touches the GPU), relays rank 0's ONE line and exits 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--backend", "gloo", "--share-device",
                          "--no-extras"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert run.returncode == 0, run.stderr[-2000:]
    d = last_json(run.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["stacks_per_step"] == 1 and d["scaling"] == "strong"
    # --no-extras: the mirror is there (the driver's record keeps `config` whole) and says that nothing BESIDE the headline was measured
    m = d["config"]["multi_gpu"]
    assert m["strong"]["value"] == d["value"] and m["strong"]["ms_per_step"] == d["ms_per_step"]
    assert m["weak"] is None and m["end_to_end"] is None and m["source_exchange_ms"] == {"broadcast": None, "all_gather": None, "bands": None}


def assert_mirrored(d: dict) -> None:
    """`config.multi_gpu` repeats the numbers of the top-level sections (which the driver's record reduces to their names) and
    `config.sharding` says which of them answers north_star's ">= 6x at 8 GPUs"."""
    m = d["config"]["multi_gpu"]
    # the number the driver divides by the N = 1 line is the fixed-total-work one: it can fall short of N
    assert d["scaling"] == "strong" and d["value"] == m["strong"]["value"] and d["ms_per_step"] == m["strong"]["ms_per_step"]
    assert d["weak"]["scaling"] == "weak" and m["weak"]["stacks_per_step"] == d["n_gpus"]
    for section in ("strong", "weak", "end_to_end", "end_to_end_bands", "end_to_end_all_gather", "field_axis_sharding"):
        assert m[section]["value"] == d[section]["value"] > 0 and m[section]["ms_per_step"] == d[section]["ms_per_step"], section
    assert m["source_exchange_ms"] == {k: d["source_exchange_ms"][k] for k in ("broadcast", "all_gather", "bands")}
    assert m["end_to_end"]["verified_bit_equal"] is True and m["end_to_end_bands"]["verified_bit_equal"] is True
    assert m["end_to_end_all_gather"]["verified_bit_equal"] is True
    assert m["secondary_timed_out_in"] is None
    text = d["config"]["sharding"]
    assert "end_to_end" in text and "weak" in text and "north_star" in text and "STRONG" in text and "FIXED TOTAL WORK" in text
    assert_wall_clock_accounted(d)


SECTIONS = ["weak", "data group", "exchange broadcast", "end_to_end", "exchange bands", "end_to_end_bands", "config4", "config5",
            "exchange all_gather", "end_to_end_all_gather", "field_axis_sharding"]


def assert_wall_clock_accounted(d: dict, skipped: bool = False) -> None:
    """The N > 1 line says where its own wall clock went — process start to the printed line, the set-up stages before `value`, every
    secondary section — and stays under the stated cap; the same figures are mirrored into `config` (which the driver's record keeps)."""
    m = d["config"]["multi_gpu"]
    assert 0 <= d["setup_s_rank"] < d["n_gpus"]  # (the set-up stages are the slowest rank's own)
    for key in ("wall_s", "setup_s", "setup_s_rank", "precompute_s", "sections_s", "sections_skipped", "total_seconds_cap"):
        assert m[key] == d[key], key
    assert m["total_seconds_cap"] == 300.0 and 0 < d["wall_s"] < m["total_seconds_cap"]  # the whole command, process start -> print
    setup = d["setup_s"]
    assert {"process_start_to_main", "library_device_and_host_group", "grids_and_knn_table", "plan_and_shard_bounds",
            "source_stacks_and_tables_resident", "timed_region", "roofline_launches"} <= set(setup)
    assert all(v >= 0 for v in setup.values()) and d["precompute_s"] <= setup["grids_and_knn_table"] + 1e-3
    if not skipped:
        assert d["sections_skipped"] == [] and list(d["sections_s"])[:len(SECTIONS)] == SECTIONS  # run in priority order, none dropped
    assert all(v >= 0 for v in d["sections_s"].values())
    # the parts add up to the whole (barriers and the JSON itself are the slack)
    assert sum(setup.values()) + sum(d["sections_s"].values()) <= d["wall_s"] + 1.0


def test_two_ranks_rehearsal_over_gloo():
    """`--gpus 2` by default carries every N > 1 line: value (fixed total work = strong scaling, exchange excluded), the weak line of
    rounds 1-4, the source exchanges timed and verified bit-equal, end_to_end (exchange inside), field-axis sharding — no opt-in flags."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--backend", "gloo", "--share-device"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-2000:]
    d = last_json(run.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["stacks_per_step"] == 1 and d["config"]["launches_per_step_per_gpu"] == 1 and d["weak"]["stacks_per_step"] == 2
    j = d["roofline"]["job"]  # both ranks' bytes against 2 x the peak
    assert j["peak"] == 16000.0 and j["algorithmic_bytes_per_step_all_ranks"] > d["roofline"]["algorithmic_bytes_per_launch"] and j["frac"] > 0
    assert "EXCLUDES the source exchange" in d["config"]["sharding"] and "gloo" in d["config"]["collectives"]
    assert set(d["source_exchange_ms"]) == {"broadcast", "all_gather", "bands"} and all(v > 0 for v in d["source_exchange_ms"].values())
    for kind in ("broadcast", "all_gather", "bands"):
        assert d["source_exchange"][kind]["verified_bit_equal"] is True
    assert d["source_exchange"]["bands"]["bytes_received_per_gpu"] < d["source_exchange"]["broadcast"]["bytes_received_per_gpu"]
    assert d["end_to_end"]["verified_bit_equal"] is True and d["end_to_end_bands"]["verified_bit_equal"] is True  # (small grids: no section skipped)
    assert d["end_to_end"]["value"] > 0 and d["end_to_end"]["value"] < d["weak"]["value"]  # the same N-stack step with the exchange inside
    assert d["strong"]["scaling"] == "strong" and d["strong"]["value"] == d["value"]
    assert d["config4"]["value"] > 0 and d["config4"]["fields"] == 24 * 7
    assert d["config5"]["value"] > 0 and d["config5"]["scaling"] == "strong"
    assert d["field_axis_sharding"]["value"] > 0  # the no-exchange comparison point rides along
    assert "cpu_baseline" not in d and "secondary_timed_out_in" not in d  # rank 0 at N = 1 only
    assert_mirrored(d)
    assert d["config"]["multi_gpu"]["config4"]["value"] == d["config4"]["value"] and d["config"]["multi_gpu"]["config5"]["value"] == d["config5"]["value"]


def test_four_ranks_rehearsal_over_gloo():
    """The same line at world 4 (four ranks sharing the test box's GPU over gloo — within the six processes a box allows on its card): the
    strong-scaling headline, the weak line with four stacks per step, every exchange verified bit-equal between four ranks — what the
    first real N = 4 run prints, minus the speed."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4", *SMALL, "--backend", "gloo", "--share-device"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-2000:]
    d = last_json(run.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 4 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["stacks_per_step"] == 1 and d["weak"]["stacks_per_step"] == 4
    for kind in ("broadcast", "all_gather", "bands"):
        assert d["source_exchange"][kind]["verified_bit_equal"] is True, kind
    assert d["end_to_end"]["verified_bit_equal"] is True and d["end_to_end_bands"]["verified_bit_equal"] is True
    assert d["end_to_end_all_gather"]["verified_bit_equal"] is True
    assert d["config4"]["value"] > 0 and d["config5"]["value"] > 0 and "secondary_timed_out_in" not in d
    assert_mirrored(d)


def test_secondary_lines_cannot_cost_the_value():
    """A stuck secondary section (budget of 0 seconds: the watchdog fires at once) still yields the ONE JSON line with `value`."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--backend", "gloo", "--share-device",
           "--secondary-seconds", "0"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert run.returncode != 0  # the line is out, but a collective that never returned is a failure the launcher must see
    d = last_json(run.stdout)
    assert d["value"] > 0 and d["n_gpus"] == 2 and "secondary_timed_out_in" in d
    assert d["config"]["multi_gpu"]["secondary_timed_out_in"] == d["secondary_timed_out_in"]  # the cut is visible in `config` too


def test_total_seconds_cap_skips_sections_and_keeps_the_line():
    """`--total-seconds` below what the process has already spent: every secondary section is skipped (all ranks agree, nobody hangs),
    the line is printed with `value`, the run ends with status 0 and says what it dropped."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--backend", "gloo", "--share-device",
           "--total-seconds", "1"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert run.returncode == 0, run.stderr[-2000:]
    d = last_json(run.stdout)
    assert d["value"] > 0 and d["n_gpus"] == 2 and "secondary_timed_out_in" not in d and d["total_seconds_cap"] == 1.0
    assert d["sections_skipped"][:4] == ["weak", "exchange broadcast", "end_to_end", "exchange bands"] and "config4" in d["sections_skipped"]
    assert "skipped" in d["weak"] and "--total-seconds" in d["weak"]["skipped"] and "skipped" in d["config5"]
    assert set(d["sections_s"]) == {"data group"}  # (bringing the group up is not a measurement: it is not skippable)
    m = d["config"]["multi_gpu"]
    assert m["strong"]["value"] == d["value"] and m["weak"] == {"skipped": d["weak"]["skipped"]} and m["sections_skipped"] == d["sections_skipped"]
    assert "value" in run.stderr and "after process start" in run.stderr  # the headline's ingredients went to stderr the moment they existed


def _two_ranks(extra_env: dict, *flags: str):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--backend", "gloo", "--share-device", *flags]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, **extra_env))


def test_a_collective_library_that_does_not_come_up_costs_only_the_exchange_sections():
    """The data group failing to come up (injected) is recorded, the sections that need no exchange still run, the line is printed, status 0."""
    run = _two_ranks({"ATX_BENCH_TEST_FAULT": "data_group"})
    assert run.returncode == 0, run.stderr[-2000:]
    d = last_json(run.stdout)
    assert d["value"] > 0 and d["weak"]["value"] > 0 and "injected" in d["source_exchange"]["error"]
    assert d["config4"]["value"] > 0 and d["config5"]["value"] > 0 and d["field_axis_sharding"]["value"] > 0
    assert "end_to_end" not in d and d["config"]["multi_gpu"]["end_to_end"] is None and d["wall_s"] > 0


def test_a_failure_outside_every_section_still_prints_the_measured_line():
    """An exception outside the per-section guards (injected): rank 0 prints `value` with `secondary_error`, every rank ends with a
    failure status — no rank goes on into collectives the failed one will never join."""
    run = _two_ranks({"ATX_BENCH_TEST_FAULT": "outside_sections"})
    assert run.returncode != 0
    d = last_json(run.stdout)
    assert d["value"] > 0 and d["n_gpus"] == 2 and "injected" in d["secondary_error"]
    assert d["config"]["multi_gpu"]["secondary_error"] == d["secondary_error"] and d["config"]["multi_gpu"]["strong"]["value"] == d["value"]


def test_multi_gpu_sections_on_real_rccl_at_world_1():
    """Everything `bench.py --gpus N` does after `value` — the nccl data group, both exchanges, end to end, and the same through
    the C-ABI communicator — on the real collective library, at the world size one GPU allows."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_PORT=str(port), MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HSA_ENABLE_IPC_MODE_LEGACY"):  # bench.py sets the IPC mode itself, before HIP starts
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--rehearse-multi"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    d = last_json(run.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0 and "secondary_timed_out_in" not in d and "nccl" in d["config"]["collectives"]
    ex = d["source_exchange"]
    assert ex["broadcast"]["verified_bit_equal"] is True and ex["bands"]["verified_bit_equal"] is True
    assert d["end_to_end"]["verified_bit_equal"] is True and d["strong"]["value"] == d["value"] and d["weak"]["value"] > 0
    assert d["end_to_end_bands"]["verified_bit_equal"] is True
    assert_mirrored(d)
    c = ex["c_abi"]
    assert c["init"]["rccl_version"] >= 20000
    for name in ("broadcast", "all_gather", "bands", "end_to_end"):
        assert c[name]["verified_bit_equal"] is True, c
    assert {"c_abi init", "c_abi broadcast", "c_abi all_gather", "c_abi bands", "c_abi end_to_end"} <= set(d["sections_s"])


def test_integration_md_ctypes_stub_runs(tmp_path):
    """The reference-side binding shown in INTEGRATION.md is executable as written (only the library path is substituted)
    and gives scipy's bits."""
    import re

    import numpy as np
    import torch
    from scipy.sparse import csr_array

    from anemoi_transform_amd import interp, native
    from anemoi_transform_amd.grids import lookup

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "class MIRMatrixHBM:" in b)
    stub = stub.replace('ctypes.CDLL("libatx.so")', f'ctypes.CDLL({native.lib_path()!r})')
    scope: dict = {}
    exec(compile(stub, "INTEGRATION.md", "exec"), scope)
    exec(compile(next(b for b in blocks if "class MIRMatrixHBMColumns" in b), "INTEGRATION.md", "exec"), scope)  # the column-stack variant

    src, tgt = lookup("o32"), lookup([5.0, 5.0])
    idx, w = interp.knn_inverse_distance(src, tgt, k=4)
    matrix = interp.ell_to_csr(idx, w, len(src["latitudes"]))
    path = str(tmp_path / "m.npz")
    np.savez(path, **matrix)
    rng = np.random.default_rng(1)

    class F:  # the two methods of an earthkit field the stub touches
        def __init__(self, values):
            self.values = values

        def to_numpy(self, flatten=False):
            return self.values

    fields = [F(280 + rng.standard_normal(len(src["latitudes"]))) for _ in range(3)]
    out = scope["MIRMatrixHBM"](path)(fields)
    torch.cuda.synchronize()
    m = csr_array((matrix["matrix_data"], matrix["matrix_indices"], matrix["matrix_indptr"]), shape=tuple(matrix["matrix_shape"]))
    for f, got in zip(fields, out.cpu().numpy()):
        assert np.array_equal(got, m @ f.values)
    assert torch.equal(scope["MIRMatrixHBMColumns"](path)(fields), out)
