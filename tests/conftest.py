import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    return O.Oracle()


@pytest.fixture(scope="session")
def gpu_lib():
    import daqp_amd
    L = daqp_amd.lib()
    if L.daqp_amd_device_count() < 1:
        pytest.fail("no HIP device visible: the -m gpu tests must run on the GPU box")
    return L
