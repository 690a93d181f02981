import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("bwd-nlkalman_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("bwd-nlkalman_amd.synth")


@pytest.fixture(scope="session")
def O():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def built(pkg):
    """The in-tree product libraries; built on demand (hipcc cross-compiles without a GPU)."""
    here = os.path.join(ROOT, "bwd-nlkalman_amd")
    if not (os.path.exists(os.path.join(here, "libnlk_hip.so"))
            and os.path.exists(os.path.join(here, "libnlkalman.so"))):
        pkg.build()
    return pkg


@pytest.fixture(scope="session")
def ctx(built):
    if built.hip().nlk_device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    c = built.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _switches_follow_the_environment(monkeypatch):
    """The product reads its NLK_* switches once per context (include/nlk_hip.h:
    nlk_ctx_reload_switches); the tests change them under the session's live contexts. So: every
    test starts from the current environment, and monkeypatch.setenv / delenv re-read it."""
    def reload_():
        pkg_ = sys.modules.get("bwd-nlkalman_amd")   # (not imported yet: no context exists either)
        if pkg_ is not None:
            pkg_.reload_switches()
    reload_()
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv

    def setenv_(name, value, *a, **k):
        setenv(name, value, *a, **k)
        reload_()

    def delenv_(name, *a, **k):
        delenv(name, *a, **k)
        reload_()
    monkeypatch.setenv, monkeypatch.delenv = setenv_, delenv_
    yield
