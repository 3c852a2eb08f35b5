import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size oracle cases (tens of seconds on 8 CPU cores)")


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(autouse=True)
def _diag_switches_follow_the_environment(request):
    """The library reads its MPA_* diagnostic switches once per process (csrc/mpa_diag.h).  Tests that pin a kernel variant
    set them with `setenv_diag` (below), which re-reads them; this fixture (set up before, hence torn down after,
    monkeypatch) re-reads them once the test's environment has been restored."""
    yield
    if "setenv_diag" in request.fixturenames:
        from multipitch_architectures_amd import _lib
        _lib.load().mpa_diag_reload()


@pytest.fixture
def setenv_diag(_diag_switches_follow_the_environment, monkeypatch):
    """setenv_diag(NAME, value) / setenv_diag(NAME, None): set / unset an MPA_* switch and make the library re-read them"""
    def _set(name, value):
        from multipitch_architectures_amd import _lib
        if value is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, str(value))
        _lib.load().mpa_diag_reload()
    return _set
