import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_engine():
    import engines

    return engines.OracleEngine()


@pytest.fixture(scope="session")
def emu_engine():
    import engines

    return engines.EmuEngine()


@pytest.fixture(scope="session")
def gpu_engine():
    import engines

    return engines.GpuEngine()
