import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture
def vmtl_env(monkeypatch):
    """set(name, value): set a VMTL_* tuning override for this test.  The library caches them after the first read
    (no getenv on the launch path), so every change is followed by vmtl_reload_env() - also when the test ends."""
    from vision_mtl_amd._lib import lib

    def set_(name, value):
        monkeypatch.setenv(name, str(value))
        lib().raw("vmtl_reload_env")()

    yield set_
    monkeypatch.undo()
    lib().raw("vmtl_reload_env")()
