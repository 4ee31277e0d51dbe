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
    """The CPU oracle (test infrastructure): ctypes bindings to oracle/liboracle.so."""
    from oracle import capi

    capi.lib()
    return capi


class EngineHooks:
    """Kernel-route options of ONE engine for the duration of a test (pmx_set_option; the library reads its environment once, at
    pmx_create, so a test that wants a route on a long-lived engine sets the option on it).  Same call shape as monkeypatch:
    setenv("PMX_SGM8_FAM", "1") / delenv("PMX_SGM8_FAM"); everything set is cleared again at the end of the test."""

    def __init__(self, eng):
        self.eng, self.saved = eng, {}

    def _remember(self, name):
        # what the option held before the test touched it (pmx_create seeds options from PMX_<name>: a run with such a variable
        # exported must get ITS route back, not "unset")
        if name not in self.saved:
            self.saved[name] = self.eng.get_option(name)

    def setenv(self, name, value):
        self._remember(name)
        self.eng.set_option(name, value)

    def delenv(self, name, raising=False):
        self._remember(name)
        self.eng.set_option(name, None)

    def undo(self):
        for n, v in self.saved.items():
            self.eng.set_option(n, v)
        self.saved.clear()


@pytest.fixture
def hooks(eng):
    h = EngineHooks(eng)
    yield h
    h.undo()
