"""`torch.ops.zsv.*`: the torch-extension layer over the C ABI (csrc/torch_binding.cpp -> libzsv_torch.so).

BASELINE.json's north_star words the binding as "a thin torch cpp_extension C-ABI layer"; the harness of this package binds
`libzsv_hip.so` with ctypes (`_lib.py`), which is what `INTEGRATION.md` shows a maintainer of the reference.  This module is the
other half: the same entry points registered with torch's dispatcher (`TORCH_LIBRARY(zsv, ...)`), for callers that want operators
instead of a foreign-function interface -- C++ / TorchScript, `torch.library` tooling, or a reference-side patch of the form

    torch.ops.zsv.conv3d(x, w, None, [1, 1, 1], [0, 1, 1])          # nn.Conv3d of resnet.py:40-45, differentiable
    torch.ops.zsv.batch_norm_relu(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, 0.1, 1e-5, True)   # resnet.py:46-49

`load()` registers the operators (idempotent) and fails loudly when the library has not been built; there is no fallback.
"""
from __future__ import annotations

import os

import torch

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libzsv_torch.so")
OPERATORS = ("version", "conv3d_fwd", "conv3d_dgrad", "conv3d_wgrad", "bn_train_fwd", "bn_train_bwd", "relu_fwd", "relu_bwd",
             "linear_fwd", "linear_dgrad", "linear_wgrad", "conv3d", "batch_norm_relu", "relu", "linear")
_loaded = False


def load():
    """Register `torch.ops.zsv.*`; returns the namespace."""
    global _loaded
    if not _loaded:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C zeroshotvideoclassification_amd/csrc` "
                               "(or `python -c 'import __graft_entry__ as g; g.build()'`)")
        torch.ops.load_library(LIB_PATH)
        _loaded = True
    return torch.ops.zsv
