import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
