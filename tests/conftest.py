import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o


@pytest.fixture(autouse=True)
def _library_env_switches_follow_the_environment():
    """libjxl_hip.so reads its debug / test switches from the environment once (jxlhip_debug_reload_env, include/jxl_hip.h);
    a test that monkeypatches one calls the reload itself -- this puts the switches back once monkeypatch has restored the
    environment (an autouse fixture is set up first, hence torn down last)."""
    yield
    from libjxl_amd import abi
    if getattr(abi, "_lib", None) is not None:
        abi._lib.jxlhip_debug_reload_env()
