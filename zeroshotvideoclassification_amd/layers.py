"""``torch.nn`` layer classes whose ``forward`` runs the gfx950 kernels.

Each class subclasses the torch layer the reference instantiates, so constructor
signatures, parameter / buffer names (``state_dict`` keys), initialisation and ``repr`` are
inherited unchanged; only ``forward`` differs: it calls ``ops`` (HIP kernels through the C
ABI) and raises on CPU tensors -- there is no fallback path.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops


class Conv3d(nn.Conv3d):
    """nn.Conv3d with dilation 1, groups 1, zero padding (all the reference uses)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.dilation != (1, 1, 1) or self.groups != 1 or self.padding_mode != "zeros" \
                or isinstance(self.padding, str):
            raise NotImplementedError("only dilation=1, groups=1, zero padding are on the hot path")

    def forward(self, x, relu: bool = False, want_stats: bool = False):
        """``want_stats=True`` returns ``(y, stats)``: BatchNorm partial statistics of ``y`` gathered in
        the kernel epilogue (None when the geometry does not provide them)."""
        return ops.conv3d(x, self.weight, self.bias, self.stride, self.padding, relu=relu, want_stats=want_stats)

    def pre_supported(self, x_shape) -> bool:
        """Can this convolution take a training-mode ``BatchNorm3d -> ReLU`` in front of it inside its own kernels?"""
        return self.bias is None and ops.conv_pre_supported(x_shape, self.weight.shape, self.stride, self.padding)

    def forward_pre(self, x, coef, want_stats: bool = False):
        """``self(relu(x * scale + shift))`` with ``coef`` from ``BatchNorm3d.deferred``: the normalised tensor is never
        written (resnet.py:46-52, the mid tensor of ``Conv2Plus1D``)."""
        return ops.conv3d_pre(x, coef, self.weight, self.stride, self.padding, want_stats=want_stats)


class BatchNorm3d(nn.BatchNorm3d):
    def forward(self, x, residual=None, relu: bool = False, stats=None, skip_link=None, defer: bool = False):
        """``defer=True`` (training mode, followed by ReLU and a convolution that ``pre_supported`` it): statistics, running
        statistics and the affine coefficients WITHOUT the normalise pass -- returns ``(x_handle, coef)`` for
        ``Conv3d.forward_pre``."""
        if x.dim() != 5:
            raise ValueError(f"expected 5D input (got {x.dim()}D input)")
        if defer:
            return ops.bn_module_deferred(x, self, stats=stats)
        return ops.bn_module_act(x, self, residual=residual, relu=relu, stats=stats, skip_link=skip_link)

    def deferred(self, x, stats=None):
        return self(x, stats=stats, defer=True)


class ReLU(nn.ReLU):
    """``inplace`` is accepted for signature compatibility; the kernel writes a fresh tensor and
    keeps only the output for backward, which is what inplace ReLU keeps too."""

    def forward(self, x):
        return ops.relu(x)


class Linear(nn.Linear):
    def forward(self, x, relu: bool = False):
        lead = x.shape[:-1]
        y = ops.linear(x.reshape(-1, x.shape[-1]), self.weight, self.bias, relu=relu)
        return y.reshape(*lead, y.shape[-1])


class MaxPool3d(nn.MaxPool3d):
    def forward(self, x):
        if self.dilation not in (1, (1, 1, 1)) or self.ceil_mode or self.return_indices:
            raise NotImplementedError("MaxPool3d: dilation/ceil_mode/return_indices are not on the hot path")
        return ops.max_pool3d(x, self.kernel_size, self.stride, self.padding)


class AdaptiveAvgPool3d(nn.AdaptiveAvgPool3d):
    """Only the global (1,1,1) form the reference builds (resnet.py:222)."""

    def forward(self, x):
        if tuple(self.output_size) != (1, 1, 1):
            raise NotImplementedError("only AdaptiveAvgPool3d((1,1,1)) is on the hot path")
        return ops.mean_pool(x).reshape(x.shape[0], x.shape[1], 1, 1, 1)
