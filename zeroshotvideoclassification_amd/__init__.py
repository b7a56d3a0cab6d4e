"""MI355X-native hot path of damien911224/ZeroShotVideoClassification.

Public surface mirrors the reference's two model files:

    from zeroshotvideoclassification_amd import network, resnet
    model = network.get_network(opt)        # network.py:24 of the reference

Everything numeric runs in ``libzsv_hip.so`` (hand-written gfx950 kernels, C ABI in
``include/zsv_hip.h``); importing the package does not need a GPU, calling an op does.
"""
from . import synthetic  # noqa: F401
from . import ddp, inference, layers, network, ops, optim, preprocess, resnet, train  # noqa: F401
from .network import C3D, MLP, Model, ResNet18, get_network  # noqa: F401

__all__ = ["network", "resnet", "ops", "layers", "train", "ddp", "optim", "inference", "preprocess", "synthetic",
           "get_network", "Model", "C3D", "MLP", "ResNet18"]
