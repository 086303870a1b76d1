import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a gfx950 (MI355X) device; run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def gpu():
    import dumphfdl_amd as hf
    if hf.device_count() < 1:
        pytest.fail("no gfx950 device visible: GPU tests must run on the GPU box (there is no CPU fallback)")
    return hf
