"""`python bench.py --gpus N` with no launcher around it (the shape of the driver's N = 1 command): the parent starts N fresh
worker processes through torch.distributed.run, relays rank 0's single line and passes the job's exit status on.  Runs without a
GPU through `--launch-check` (every rank joins the host-side gloo group; nothing touches the device)."""

from __future__ import annotations

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                              "HSA_ENABLE_IPC_MODE_LEGACY")}


def test_self_launch_relays_one_line_and_sets_the_ipc_mode_before_hip_starts():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"], capture_output=True, text=True,
                         timeout=300, cwd=ROOT, env=_env())
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, run.stdout  # the collective library's connection notes went to stderr
    d = json.loads(lines[0])
    assert d == {"launch_check": True, "n_gpus": 3, "max_rank_seen": 2, "HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def test_self_launch_passes_a_failure_on():
    """Without a GPU the real bench fails in every worker: the parent must exit non-zero and print no line."""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("needs a box without a GPU")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--src-grid", "o16", "--tgt-grid", "10.0", "--levels", "3",
                          "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=_env())
    assert run.returncode != 0 and not run.stdout.strip()


def test_mismatched_world_size_is_refused():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                         timeout=120, cwd=ROOT, env=dict(_env(), WORLD_SIZE="4", RANK="0"))
    assert run.returncode != 0 and "does not match WORLD_SIZE" in run.stderr
