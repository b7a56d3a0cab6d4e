"""MI355X-native video ResNets behind the reference's ``resnet.py`` surface.

Drop-in for the reference module of the same name (``import resnet as models``,
network.py:5): identical public names, constructor signatures, attribute tree and
``state_dict`` keys; ``VideoResNet.forward`` returns ``(pooled, layer4_features)`` like the
fork does (resnet.py:243-256).  All arithmetic runs in the hand-written gfx950 kernels
(``ops``): the containers below do not call their children one by one -- they launch the
fused sequences the hardware wants:

    conv (MFMA implicit GEMM)  ->  BatchNorm statistics  ->  normalise + residual + ReLU
                                                             in a single HBM pass

* ``Conv2Plus1D``        resnet.py:37-57    spatial 1x3x3 -> BN+ReLU (fused) -> temporal 3x1x1
* ``BasicBlock``         resnet.py:79-113   ... -> BN + ``out += residual`` + ReLU fused
* ``R2Plus1dStem`` / ``BasicStem``  resnet.py:165-187
* ``VideoResNet``        resnet.py:190-281, factories resnet.py:293-362
"""
from __future__ import annotations

from typing import Any, Callable, List, Optional, Sequence, Tuple, Type, Union

import torch
from torch import Tensor, nn

import os

from . import ops
from .layers import AdaptiveAvgPool3d, BatchNorm3d, Conv3d, Linear, ReLU

__all__ = ["r3d_18", "mc3_18", "r2plus1d_18"]


def _has_hooks(*mods: nn.Module) -> bool:
    """A forward (pre-)hook on one of the modules of a `BatchNorm -> ReLU -> conv` triple must see what the reference's module
    would hand it (the normalised tensor, the activated tensor): such a triple takes the separate passes, not the fold."""
    return any(m._forward_hooks or m._forward_pre_hooks for m in mods)


def _call(m: nn.Module, x: Tensor, want_stats: bool):
    """Run a conv-like child; with ``want_stats`` also fetch the BatchNorm partial statistics its
    (last) convolution accumulated in the kernel epilogue."""
    if want_stats and isinstance(m, (Conv3d, _FusedSequential)) and not _has_hooks(m):
        return m(x, want_stats=True)
    return m(x), None                  # (a hooked module is called the plain way: its hooks see a tensor, not (y, statistics))


def _run_chain(mods: Sequence[nn.Module], x: Tensor, want_stats: bool = False):
    """Run conv / BN / ReLU children as fused kernel sequences: a convolution followed by a
    training-mode BatchNorm hands over its epilogue statistics (no separate pass over the
    activations), and BN + ReLU are one pass.  Returns ``(x, stats_of_last_conv_or_None)``."""
    mods = list(mods)
    i, n = 0, len(mods)
    stats = None
    while i < n:
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < n else None
        if isinstance(m, BatchNorm3d):
            relu = isinstance(nxt, nn.ReLU) and not _has_hooks(m, nxt)      # hooked: BatchNorm and ReLU run (and are observed) one by one
            after = mods[i + 2] if (relu and i + 2 < n) else None
            if relu and m.training and m.momentum is not None and type(after) is Conv3d and x.is_contiguous() \
                    and not _has_hooks(m, nxt, after) and after.pre_supported(x.shape):
                # BN -> ReLU -> conv (Conv2Plus1D's mid tensor, resnet.py:46-52): the convolution applies the affine +
                # ReLU while it reads the raw tensor; the normalised tensor is never written or read back
                handle, coef = m(x, stats=stats, defer=True)
                after_next = mods[i + 3] if i + 3 < n else None
                feeds = isinstance(after_next, BatchNorm3d) and after_next.training
                x, stats = after.forward_pre(handle, coef, want_stats=True) if (feeds or (i + 3 == n and want_stats)) \
                    else (after.forward_pre(handle, coef), None)
                i += 3
                continue
            x = m(x, relu=relu, stats=stats)
            stats = None
            i += 2 if relu else 1
            continue
        feeds_bn = isinstance(nxt, BatchNorm3d) and nxt.training
        last = i + 1 == n
        x, stats = _call(m, x, feeds_bn or (last and want_stats))
        i += 1
    return x, stats


class _FusedSequential(nn.Sequential):
    def forward(self, x: Tensor, want_stats: bool = False):
        x, stats = _run_chain(list(self), x, want_stats)
        return (x, stats) if want_stats else x


class Conv3DSimple(Conv3d):
    """3x3x3 convolution, stride s in every axis (resnet.py:18-34)."""

    def __init__(self, in_planes: int, out_planes: int, midplanes: Optional[int] = None, stride: int = 1,
                 padding: int = 1) -> None:
        super().__init__(in_planes, out_planes, kernel_size=(3, 3, 3), stride=stride, padding=padding, bias=False)

    @staticmethod
    def get_downsample_stride(stride: int) -> Tuple[int, int, int]:
        return stride, stride, stride


class Conv3DNoTemporal(Conv3d):
    """1x3x3 convolution, spatial stride only (resnet.py:60-76)."""

    def __init__(self, in_planes: int, out_planes: int, midplanes: Optional[int] = None, stride: int = 1,
                 padding: int = 1) -> None:
        super().__init__(in_planes, out_planes, kernel_size=(1, 3, 3), stride=(1, stride, stride),
                         padding=(0, padding, padding), bias=False)

    @staticmethod
    def get_downsample_stride(stride: int) -> Tuple[int, int, int]:
        return 1, stride, stride


class Conv2Plus1D(_FusedSequential):
    """Factorised (2+1)D convolution (resnet.py:37-57): children 0..3 are the spatial conv, its
    BatchNorm, ReLU, and the temporal conv -- same indices, hence same ``state_dict`` keys."""

    def __init__(self, in_planes: int, out_planes: int, midplanes: int, stride: int = 1, padding: int = 1) -> None:
        super().__init__(
            Conv3d(in_planes, midplanes, kernel_size=(1, 3, 3), stride=(1, stride, stride),
                   padding=(0, padding, padding), bias=False),
            BatchNorm3d(midplanes),
            ReLU(inplace=True),
            Conv3d(midplanes, out_planes, kernel_size=(3, 1, 1), stride=(stride, 1, 1),
                   padding=(padding, 0, 0), bias=False),
        )

    @staticmethod
    def get_downsample_stride(stride: int) -> Tuple[int, int, int]:
        return stride, stride, stride


def _midplanes(inplanes: int, planes: int) -> int:
    # parameter-matching width of the factorised pair (resnet.py:91)
    return (inplanes * planes * 3 * 3 * 3) // (inplanes * 3 * 3 + 3 * planes)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes: int, planes: int, conv_builder: Callable[..., nn.Module], stride: int = 1,
                 downsample: Optional[nn.Module] = None) -> None:
        super().__init__()
        mid = _midplanes(inplanes, planes)
        self.conv1 = _FusedSequential(conv_builder(inplanes, planes, mid, stride), BatchNorm3d(planes),
                                      ReLU(inplace=True))
        self.conv2 = _FusedSequential(conv_builder(planes, planes, mid), BatchNorm3d(planes))
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x: Tensor) -> Tensor:
        tail = self.conv2[1] if len(self.conv2) == 2 else None
        fused_tail = isinstance(tail, BatchNorm3d)
        # identity shortcut: x feeds conv1's first convolution AND the tail's `out += residual`; link the
        # two so that the shortcut's gradient is added inside that convolution's dgrad (ops.SkipLink)
        link = None
        if self.downsample is None and fused_tail and torch.is_grad_enabled() and x.requires_grad \
                and os.environ.get("ZSV_NO_SKIP_FUSION") is None:
            link = ops.SkipLink()
            x._zsv_skip_link = link
        # strided 1x1x1 shortcut convolution: x feeds conv1's first (strided) convolution AND `downsample`; link the two so that
        # the shortcut's input gradient is added inside that convolution's dgrad in compact form (ops.DownLink)
        down = None
        ds_conv = self.downsample[0] if isinstance(self.downsample, nn.Sequential) and len(self.downsample) > 0 else None
        if isinstance(ds_conv, Conv3d) and tuple(ds_conv.kernel_size) == (1, 1, 1) and torch.is_grad_enabled() and x.requires_grad \
                and os.environ.get("ZSV_NO_DOWN_FUSION") is None:
            down = ops.DownLink(ds_conv.stride)
            x._zsv_down_link = down
        out = self.conv1(x)
        if link is not None:
            x.__dict__.pop("_zsv_skip_link", None)           # (not consumed: conv1 does not start with a Conv3d)
        if down is not None:
            x.__dict__.pop("_zsv_down_link", None)
        out, stats = _call(self.conv2[0], out, fused_tail and tail.training)
        if down is not None and down.armed:
            x._zsv_down_src = down
        residual = x if self.downsample is None else self.downsample(x)
        if down is not None:
            x.__dict__.pop("_zsv_down_src", None)
        if fused_tail:
            # BN + `out += residual` + ReLU (resnet.py:97,110-111) in one pass
            return tail(out, residual=residual, relu=True, stats=stats, skip_link=link)
        out, _ = _run_chain(list(self.conv2)[1:], out)
        return ops.add_relu(out, residual)


class Bottleneck(nn.Module):
    """Present for surface completeness (resnet.py:116-162); no factory in the reference uses it."""
    expansion = 4

    def __init__(self, inplanes: int, planes: int, conv_builder: Callable[..., nn.Module], stride: int = 1,
                 downsample: Optional[nn.Module] = None) -> None:
        super().__init__()
        mid = _midplanes(inplanes, planes)
        self.conv1 = _FusedSequential(Conv3d(inplanes, planes, kernel_size=1, bias=False), BatchNorm3d(planes),
                                      ReLU(inplace=True))
        self.conv2 = _FusedSequential(conv_builder(planes, planes, mid, stride), BatchNorm3d(planes),
                                      ReLU(inplace=True))
        self.conv3 = _FusedSequential(Conv3d(planes, planes * self.expansion, kernel_size=1, bias=False),
                                      BatchNorm3d(planes * self.expansion))
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x: Tensor) -> Tensor:
        out = self.conv2(self.conv1(x))
        out, stats = _call(self.conv3[0], out, self.conv3[1].training)
        residual = x if self.downsample is None else self.downsample(x)
        return self.conv3[1](out, residual=residual, relu=True, stats=stats)


class BasicStem(_FusedSequential):
    """conv 3x7x7 / (1,2,2) -> BN -> ReLU (resnet.py:165-173)."""

    def __init__(self) -> None:
        super().__init__(
            Conv3d(3, 64, kernel_size=(3, 7, 7), stride=(1, 2, 2), padding=(1, 3, 3), bias=False),
            BatchNorm3d(64), ReLU(inplace=True))


class R2Plus1dStem(_FusedSequential):
    """conv 1x7x7 / (1,2,2) -> BN -> ReLU -> conv 3x1x1 -> BN -> ReLU (resnet.py:176-187)."""

    def __init__(self) -> None:
        super().__init__(
            Conv3d(3, 45, kernel_size=(1, 7, 7), stride=(1, 2, 2), padding=(0, 3, 3), bias=False),
            BatchNorm3d(45), ReLU(inplace=True),
            Conv3d(45, 64, kernel_size=(3, 1, 1), stride=(1, 1, 1), padding=(1, 0, 0), bias=False),
            BatchNorm3d(64), ReLU(inplace=True))


class VideoResNet(nn.Module):
    def __init__(self, block: Type[Union[BasicBlock, Bottleneck]],
                 conv_makers: Sequence[Type[Union[Conv3DSimple, Conv3DNoTemporal, Conv2Plus1D]]],
                 layers: List[int], stem: Callable[..., nn.Module], num_classes: int = 400,
                 zero_init_residual: bool = False) -> None:
        super().__init__()
        self.inplanes = 64
        self.stem = stem()
        for i, (planes, stride) in enumerate([(64, 1), (128, 2), (256, 2), (512, 2)]):   # resnet.py:217-220
            setattr(self, f"layer{i + 1}", self._make_layer(block, conv_makers[i], planes, layers[i], stride=stride))
        self.avgpool = AdaptiveAvgPool3d((1, 1, 1))
        self.fc = Linear(512 * block.expansion, num_classes)      # built but never applied (resnet.py:254)
        self._init_weights(zero_init_residual)

    def _init_weights(self, zero_init_residual: bool) -> None:
        # resnet.py:226-241
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.zeros_(m.conv3[1].weight)

    def forward(self, x: Tensor):
        with ops.batched_bn_counters():
            x = self.stem(x)
            x = self.layer1(x)
            x = self.layer2(x)
            x = self.layer3(x)
            f = self.layer4(x)
        pooled = ops.mean_pool(f)          # avgpool + flatten(1); `fc` is skipped by the fork
        return pooled, f

    def _make_layer(self, block, conv_builder, planes: int, blocks: int, stride: int = 1) -> nn.Sequential:
        downsample = None
        out_planes = planes * block.expansion
        if stride != 1 or self.inplanes != out_planes:                      # resnet.py:268-273
            downsample = _FusedSequential(
                Conv3d(self.inplanes, out_planes, kernel_size=1, stride=conv_builder.get_downsample_stride(stride),
                       bias=False),
                BatchNorm3d(out_planes))
        stack = [block(self.inplanes, planes, conv_builder, stride, downsample)]
        self.inplanes = out_planes
        stack += [block(self.inplanes, planes, conv_builder) for _ in range(1, blocks)]
        return nn.Sequential(*stack)


def _video_resnet(arch: str, pretrained: bool = False, progress: bool = True, **kwargs: Any) -> VideoResNet:
    if pretrained:
        # the reference would download Kinetics weights here (resnet.py:287-289); its CLI can never
        # request that and there is no network
        raise RuntimeError(f"pretrained weights for {arch} are not available offline; load a checkpoint with "
                           "load_state_dict instead")
    return VideoResNet(**kwargs)


def r3d_18(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> VideoResNet:
    """R3D-18: 3x3x3 convolutions throughout (resnet.py:293-314)."""
    return _video_resnet("r3d_18", pretrained, progress, block=BasicBlock, conv_makers=[Conv3DSimple] * 4,
                         layers=[2, 2, 2, 2], stem=BasicStem, **kwargs)


def mc3_18(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> VideoResNet:
    """MC3-18: 3-D first stage, 2-D (1x3x3) afterwards (resnet.py:318-338)."""
    return _video_resnet("mc3_18", pretrained, progress, block=BasicBlock,
                         conv_makers=[Conv3DSimple] + [Conv3DNoTemporal] * 3, layers=[2, 2, 2, 2], stem=BasicStem,
                         **kwargs)


def r2plus1d_18(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> VideoResNet:
    """R(2+1)D-18: every 3-D convolution factorised into 1x3x3 + 3x1x1 (resnet.py:342-362)."""
    return _video_resnet("r2plus1d_18", pretrained, progress, block=BasicBlock, conv_makers=[Conv2Plus1D] * 4,
                         layers=[2, 2, 2, 2], stem=R2Plus1dStem, **kwargs)
