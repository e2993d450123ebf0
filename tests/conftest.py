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


@pytest.fixture(autouse=True)
def _cs_switches(monkeypatch):
    """The library reads its CS_* switches from the environment ONCE (custrings_amd/csrc/cs_config.h) and changes them at run
    time only through cs_config_set: a test that sets one with monkeypatch.setenv tells the library too, and the switch goes
    back to what the process started with afterwards."""
    touched = {}
    plain_setenv, plain_delenv = monkeypatch.setenv, monkeypatch.delenv

    def tell(name, value):
        # (a CPU-only test -- rowemu, oracle -- may touch a CS_ name on a box without the GPU runtime: the library is then
        # simply not there to be told)
        try:
            from custrings_amd import _lib
        except (ImportError, OSError):
            return
        _lib.lib.cs_config_set(name.encode(), None if value is None else str(value).encode())

    def setenv(name, value, prepend=None):
        if name.startswith("CS_") and name not in touched:
            touched[name] = os.environ.get(name)
        plain_setenv(name, value, prepend)
        if name.startswith("CS_"):
            tell(name, value)

    def delenv(name, raising=True):
        if name.startswith("CS_") and name not in touched:
            touched[name] = os.environ.get(name)
        plain_delenv(name, raising)
        if name.startswith("CS_"):
            tell(name, None)

    monkeypatch.setenv = setenv
    monkeypatch.delenv = delenv
    yield
    for name, before in touched.items():
        tell(name, before)


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
