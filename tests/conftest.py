"""Shared pytest setup: import the in-tree package, register the ``gpu`` marker."""

from __future__ import annotations

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

if not os.path.exists(graft.LIB):  # fresh checkout: build libatx.so (hipcc cross-compiles without a GPU)
    graft.build()
graft.load_package()

# the k-NN table files of interp.py go to a directory of this test session, not to the user's cache
import atexit  # noqa: E402
import shutil  # noqa: E402
import tempfile  # noqa: E402

if "ATX_CACHE_DIR" not in os.environ:
    _cache = tempfile.mkdtemp(prefix="atx-test-cache-")
    os.environ["ATX_CACHE_DIR"] = _cache
    atexit.register(shutil.rmtree, _cache, True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. a plain `pytest tests/` on the CPU box."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch

    assert torch.cuda.is_available()
    d = torch.device("cuda", 0)
    torch.cuda.set_device(d)
    return d
