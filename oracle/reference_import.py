"""Import the reference's ``resnet.py`` / ``network.py`` in the build container.

TEST INFRASTRUCTURE.  Used only by ``oracle/make_golden.py`` and the container-only
tests that pin ``oracle/restatement.py`` to the reference.  ``/root/reference`` does
not exist on the GPU box; everything there works from ``tests/golden``.

The two files need a few symbols that are absent offline and carry no arithmetic
(SURVEY.md section 8c): ``torchvision._internally_replaced_utils.load_state_dict_from_url``
(only reached with ``pretrained=True``; never),
``torchvision.utils._log_api_usage_once`` (telemetry no-op), ``gensim.models.KeyedVectors``
and ``clip`` (imported by ``network.py:8,19``; no live code touches them).  They are
provided as in-memory stub modules; nothing is fetched and no bytecode is written.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("ZSV_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "network.py"))


def _stub(name: str, **attrs) -> types.ModuleType:
    mod = sys.modules.get(name)
    if mod is None:
        mod = types.ModuleType(name)
        mod.__dict__["__zsv_stub__"] = True
        sys.modules[name] = mod
    for k, v in attrs.items():
        if not hasattr(mod, k):
            setattr(mod, k, v)
    return mod


def _no_download(*_a, **_k):
    raise RuntimeError("offline: pretrained weights are never loaded (SURVEY F3)")


def import_reference():
    """Returns the reference's ``(network, resnet)`` modules."""
    if not reference_available():
        raise FileNotFoundError(REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    try:
        importlib.import_module("torchvision")
    except Exception:
        tv = _stub("torchvision")
        tv._internally_replaced_utils = _stub(
            "torchvision._internally_replaced_utils", load_state_dict_from_url=_no_download)
        tv.utils = _stub("torchvision.utils", _log_api_usage_once=lambda *_a, **_k: None)
    try:
        importlib.import_module("gensim.models")
    except Exception:
        gs = _stub("gensim")
        gs.models = _stub("gensim.models", KeyedVectors=type("KeyedVectors", (), {}))
    try:
        importlib.import_module("clip")
    except Exception:
        _stub("clip")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # the reference's files are top-level modules called ``resnet`` / ``network``
    for name in ("resnet", "network"):
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, "__file__", "").startswith(REFERENCE_ROOT):
            del sys.modules[name]
    resnet = importlib.import_module("resnet")
    network = importlib.import_module("network")
    return network, resnet
