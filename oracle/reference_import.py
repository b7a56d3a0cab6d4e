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


class _Compose:                                           # torchvision.transforms.Compose: apply in order
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def _ensure_torchvision():
    """The real ``torchvision`` when it imports; otherwise ONE stub package carrying every arithmetic-free symbol either
    importer needs (``resnet.py:6-7``: ``_internally_replaced_utils.load_state_dict_from_url``, ``utils._log_api_usage_once``;
    ``auxiliary/transforms.py:56``: ``transforms.Compose``).  Both importers call this, so their order does not matter (a
    stub left by the first call used to satisfy the second call's ``import torchvision`` while lacking its sub-modules)."""
    existing = sys.modules.get("torchvision")
    if existing is None or not getattr(existing, "__zsv_stub__", False):
        try:
            importlib.import_module("torchvision")
            importlib.import_module("torchvision.transforms")
            return
        except Exception:
            for name in [n for n in sys.modules if n == "torchvision" or n.startswith("torchvision.")]:
                del sys.modules[name]                    # a half-imported package must not shadow the stub
    tv = _stub("torchvision")
    tv.__dict__.setdefault("__path__", [])               # a package: `from torchvision.x import y` resolves through sys.modules
    tv._internally_replaced_utils = _stub("torchvision._internally_replaced_utils", load_state_dict_from_url=_no_download)
    tv.utils = _stub("torchvision.utils", _log_api_usage_once=lambda *_a, **_k: None)
    tv.transforms = _stub("torchvision.transforms", Compose=_Compose)


def import_reference():
    """Returns the reference's ``(network, resnet)`` modules."""
    if not reference_available():
        raise FileNotFoundError(REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    _ensure_torchvision()
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


def import_reference_transforms():
    """The reference's ``auxiliary/transforms.py`` (build container only).

    It imports ``imageio`` (used by ``batch2gif`` only, transforms.py:70-77) and
    ``torchvision.transforms`` (only ``Compose``, transforms.py:56); neither carries arithmetic of
    the clip chain, so an empty ``imageio`` module and a ``Compose`` that applies its list in order
    stand in for them when the packages are absent.  Every function that touches pixel values
    (``to_normalized_float_tensor``, ``resize``, ``crop``, ``center_crop``, ``RandomCrop``,
    ``RandomHorizontalFlip``) is the reference's own code."""
    if not reference_available():
        raise FileNotFoundError(REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    try:
        importlib.import_module("imageio")
    except Exception:
        _stub("imageio")

    _ensure_torchvision()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    mod = sys.modules.get("auxiliary.transforms")
    if mod is not None and not getattr(mod, "__file__", "").startswith(REFERENCE_ROOT):
        del sys.modules["auxiliary.transforms"]
    return importlib.import_module("auxiliary.transforms")


def reference_main_functions(names=("evaluate", "compute_accuracy"), **globals_):
    """Compile single functions of the reference's ``main.py`` WITHOUT running the module.

    ``main.py`` cannot be imported (module-level argparse, ``torch.device('cuda')`` and the dataset
    build, main.py:55-111), but ``evaluate`` (main.py:224-313) and ``compute_accuracy``
    (main.py:316-325) are self-contained given their globals.  The file is parsed with ``ast``,
    the named ``FunctionDef`` nodes are compiled from the reference's own text and executed in a
    namespace holding ``globals_`` (the caller supplies ``model``, ``opt``, ``Fore`` ... stand-ins
    plus the real ``np``, ``torch``, ``cdist``, ``accuracy_score``).  Nothing of the text is stored."""
    import ast
    path = os.path.join(REFERENCE_ROOT, "main.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    if sorted(n.name for n in picked) != sorted(names):
        raise RuntimeError(f"main.py does not define {names}")
    ns = dict(globals_)
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return ns
