"""The C-ABI communicator's MULTI-PEER paths at world sizes 2, 3 and 4 on the one GPU of the test box (four ranks + the test process, which holds
a GPU context of its own in a full `-m gpu` run: five of the six processes a box allows on its card).

Real RCCL needs one GPU per rank, so tests/test_gpu_rccl.py stops at world 1, where `atx_exchange` only copies its own slab and
`atx_gather_shards` broadcasts one range.  Here the ranks share the GPU and libatx binds (through ATX_RCCL_LIBRARY) the host-staged
stand-in of tests/rccl_stub — test infrastructure that moves the bytes it is told to move and REFUSES a receive whose size differs
from what the peer sent — so the per-peer send / recv bookkeeping, the shard-gather ranges, the broadcasts from every root and the
double-buffered step run for real, with the real regrid kernels, against the oracle (tests/rccl_stub_child.py)."""

from __future__ import annotations

import json
import os
import subprocess
import sys
import uuid

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
STUB_DIR = os.path.join(HERE, "rccl_stub")
STUB = os.path.join(STUB_DIR, "librccl_stub.so")
EXPECTED = ["exchange_stacks", "all_gather", "exchange_source_bands", "pipelined_sharded_regrid", "gather_target_shards", "raw_exchange"]


@pytest.fixture(scope="module")
def stub():
    src = os.path.join(STUB_DIR, "rccl_stub.cpp")
    if not os.path.exists(STUB) or os.path.getmtime(STUB) < os.path.getmtime(src):
        hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
        subprocess.run([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-o", STUB, src, "-lpthread"], check=True, cwd=STUB_DIR)
    return STUB


@pytest.mark.parametrize("world", [2, 3, 4])
def test_multi_peer_paths_of_the_c_abi_communicator(stub, world):
    token = uuid.uuid4().hex[:16]
    env = dict(os.environ, ATX_RCCL_LIBRARY=stub, HSA_ENABLE_IPC_MODE_LEGACY="0")
    children = [subprocess.Popen([sys.executable, os.path.join(HERE, "rccl_stub_child.py"), str(r), str(world), token], stdout=subprocess.PIPE,
                                 stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    try:
        results = [c.communicate(timeout=600) for c in children]
    finally:
        for c in children:
            if c.poll() is None:
                c.kill()
    for r, (c, (out, err)) in enumerate(zip(children, results)):
        assert c.returncode == 0, f"rank {r}: {err[-3000:]}"
        verdict = json.loads(out.strip().splitlines()[-1])
        assert verdict["rank"] == r and verdict["rccl_version"] == 29999  # the stand-in, not the real library
        for name in EXPECTED:
            assert verdict.get(name) is True, (r, name, verdict)
        assert verdict["ok"] is True
