import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
sys.dont_write_bytecode = True  # never drop __pycache__ into /root/reference

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain ``pytest tests`` on a machine without a GPU skips the gpu-marked tests instead of erroring."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a GPU (run on the MI355X box with -m gpu)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_l2(a, b):
    """||a-b|| / ||b|| in float64."""
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    d = (a - b).norm()
    n = b.norm()
    return float(d / n) if float(n) > 0 else float(d)


@pytest.fixture(autouse=True)
def _reset_instrumentation():
    """A test that fails between ``shadow_check_begin()`` / ``gate_log_begin()`` and their ``_end()`` must not leave the
    instrumentation switched on for the tests after it (a shadow check inside a later hipGraph capture is illegal)."""
    yield
    ops = sys.modules.get("rave_amd.ops")
    if ops is not None:
        for name in ("_SHADOW", "_PLAN_LOG", "_GATE_LOG", "_PROFILE"):
            if hasattr(ops, name):
                setattr(ops, name, None)
