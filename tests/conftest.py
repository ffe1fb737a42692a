import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def _gpu_available():
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run on the MI355X box with -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def tuning_build():
    """The test runs against libhydrochrono_amd_tuning.so (the same sources with -DHC_TUNING, hydrochrono_amd/build.py): the sweep /
    A-B / fault-injection switches it sets in the environment (HC_SUB_BLOCK, HC_MINI_NARROW, HC_WIDE_FUSED, HC_SLOT_STATE,
    HC_FAULT_STALE_STATE_AT, HC_PASS_AHEAD_MIN_MB, ...) and the kernel variants that were measured and not taken exist there only;
    the release library reads none of them.  Objects created inside the test keep that library for their lifetime."""
    from hydrochrono_amd import capi
    with capi.use_flavor("tuning"):
        yield
