import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container (GPU tests run via gpurun)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# The library snapshots its ZSV_* switches (csrc/knobs.h); `_lib.reload_knobs()` refreshes the snapshot after an os.environ
# write.  The two monkeypatch methods call it, and every test starts from a fresh snapshot (monkeypatch's undo does not pass
# through them).
def _refresh_knobs():
    try:
        from zeroshotvideoclassification_amd import _lib
        if os.path.isfile(_lib.LIB_PATH):
            _lib.reload_knobs()
    except Exception:
        pass


def _wrap(method):
    def inner(self, name, *args, **kwargs):
        out = method(self, name, *args, **kwargs)
        if str(name).startswith("ZSV_"):
            _refresh_knobs()
        return out
    return inner


pytest.MonkeyPatch.setenv = _wrap(pytest.MonkeyPatch.setenv)
pytest.MonkeyPatch.delenv = _wrap(pytest.MonkeyPatch.delenv)


@pytest.fixture(autouse=True)
def _fresh_knob_snapshot():
    _refresh_knobs()
    yield
