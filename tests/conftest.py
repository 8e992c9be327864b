import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _b2m_env_switches(monkeypatch):
    """The library reads its B2M_* switches once per process (b2m_reload_env re-reads them): a test that flips one through
    `monkeypatch` must take effect at once, and nothing may leak into the next test."""
    from box2mask_amd import _lib

    def reload():
        try:
            _lib.reload_env()
        except (ImportError, OSError, AttributeError):      # (no library, or a stale one without b2m_reload_env)
            pass
    set0, del0 = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        set0(name, value, prepend)
        reload()

    def delenv(name, raising=True):
        del0(name, raising)
        reload()
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    reload()
    yield
    # (this runs before monkeypatch restores the environment: the next test's set-up reloads again)
    reload()
