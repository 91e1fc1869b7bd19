"""The multi-GPU transport on real RCCL: a FRESH child process per transport — torch.distributed
``nccl`` with ``device_id`` (what bench.py and the driver's N > 1 runs use) and the C-ABI communicator ``atx_comm_*`` —
runs every function of anemoi_transform_amd.distributed at the world size one MI355X allows (1) and checks the results
against the oracle (tests/rccl_child.py).  World-2 semantics of the same functions: tests/test_distributed_gloo.py."""

from __future__ import annotations

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

EXPECTED = ["broadcast_stack", "exchange_stacks", "all_gather", "exchange_source_bands", "pipelined_sharded_regrid", "pipelined_repeat_10",
            "gather_target_shards", "p2p_self"]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("transport", ["torch", "atx"])
def test_source_exchange_on_rccl_world1(transport):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    child = subprocess.run([sys.executable, os.path.join(HERE, "rccl_child.py"), transport, str(_free_port())],
                           capture_output=True, text=True, timeout=600, env=env)
    assert child.returncode == 0, child.stderr[-3000:]
    verdict = json.loads(child.stdout.strip().splitlines()[-1])
    print(verdict)
    for name in EXPECTED:
        assert verdict.get(name) is True, (name, verdict)
    if transport == "torch":
        assert verdict["backend"] == "nccl" and verdict["async_broadcast_then_kernel"] is True
    else:
        assert verdict["rccl_version"] >= 20000
    assert verdict["ok"] is True
