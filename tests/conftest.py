import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    from alfalfa_amd import ffi
    if ffi.lib().alfa_device_count() <= 0:
        pytest.fail("no HIP device visible but a gpu-marked test was selected (libalfa_hot has no CPU fallback)")
    return 0
