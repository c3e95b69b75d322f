import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_usable():
    """A HIP device and the built product library (GPU tests never fall back to anything else)."""
    try:
        import torch
        if not torch.cuda.is_available():
            return False, "no HIP device"
    except Exception as e:  # pragma: no cover
        return False, "torch unavailable: %s" % e
    if not os.path.exists(os.path.join(ROOT, "libxaac_amd", "libxaac_amd.so")):
        return False, "libxaac_amd.so not built"
    return True, ""


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a CPU-only box skips the gpu-marked tests instead of erroring in their fixtures.
    (With `-m gpu` on a box that should have a GPU the tests still run and fail loudly if it is missing.)"""
    if any(it.get_closest_marker("gpu") for it in items):
        ok, why = _gpu_usable()
        if not ok and "gpu" not in (config.getoption("-m") or ""):
            skip = pytest.mark.skip(reason="gpu test: " + why)
            for it in items:
                if it.get_closest_marker("gpu"):
                    it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement (oracle/liboracle.so), built on demand with gcc."""
    import oracle_lib
    return oracle_lib.load_oracle()


@pytest.fixture(scope="session")
def reference():
    """The compiled reference via oracle/_ref/libref_harness.so; skipped where it was never built
    (it can only be built where /root/reference exists)."""
    import oracle_lib
    ref = oracle_lib.load_reference()
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return ref
