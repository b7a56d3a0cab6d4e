"""Mixed-precision (bf16) TRAINING forward / backward of the VideoResNet trunks -- the reference's
``with autocast(): Y, _ = model(X)`` (main.py:172) with ``GradScaler`` around it (main.py:137,195-203).

    with amp.autocast():                         # torch.cuda.amp.autocast's place in main.py:172
        y, _ = model(x)                          # network.Model over r2plus1d_18 / r3d_18 / mc3_18
    loss = criterion(y, z)
    scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()      # optim.LossScaler

What autocast does to the reference's trunk, restated for this chip: every ``Conv3d`` multiplies reduced-precision
operands and accumulates in fp32; ``BatchNorm3d`` keeps fp32 statistics and affine parameters on a reduced-precision
input; ReLU / ``out += residual`` are reduced-precision element-wise passes; parameters and their gradients stay fp32;
``MSELoss`` runs in fp32.  Here the reduced precision is bf16 (the MI355X matrix core's native 16-bit type; no loss
scaling is needed for its exponent range, the scaler is supported all the same):

* activations travel channels-last bf16 ``[N][T][H][W][Cp]`` (``csrc/conv_bf16.hip``'s layout);
* forward convolution: ``zsv_conv3d_bf16_fwd`` (v_mfma_f32_16x16x32_bf16, fp32 accumulation) on the fp32 master weights
  packed to bf16 once per step; train-mode BatchNorm + ReLU + residual: ``zsv_bn_cl_fwd_train`` (csrc/train_bf16.hip);
* input gradient: the same convolution kernel on the transposed, tap-flipped weights (a strided convolution's gradient is
  the stride-1 convolution of the zero-interleaved output gradient); BatchNorm / ReLU backward: ``zsv_bn_cl_bwd``;
* weight gradient: fp32 accumulation of the bf16 operands.  The stride-1 "same" convolutions (1x3x3, 3x1x1, 3x3x3) run
  ``zsv_conv3d_bf16_wgrad`` (csrc/wgrad_bf16.hip: the contraction runs over voxels, the operands are staged as they lie in
  memory and read back through gfx950's transposed LDS read); strided convolutions, the 1x1x1 shortcuts and the clip
  convolution convert the two operands to fp32 NCDHW (``zsv_cl_bf16_to_ncs_f32``) and use the fp32 kernels of the main path;
* the pooled 512-d feature, the MLP head, normalisation and the loss stay fp32 (``ops``), as under autocast's fp32 list.

The whole trunk is ONE ``torch.autograd.Function`` (its forward keeps its own tape): autograd sees
``pooled = f(clips, *trunk parameters)``.  Eval mode under ``autocast`` runs ``inference.Bf16Engine``.
There is no CPU fallback.
"""
from __future__ import annotations

import os
import threading
from ctypes import byref, c_void_p
from typing import List, Optional

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib, ops
from ._lib import ConvDesc
from .inference import _conv_bn_relu_chain, channel_pitch, clip_to_bf16, conv_bf16, meanpool_bf16, pack_conv

_state = threading.local()


class autocast:
    """``torch.cuda.amp.autocast``'s role at main.py:172 for this package's models: inside the context a training-mode
    ``network.Model`` over a VideoResNet trunk runs its trunk in bf16 (this module); in eval mode it runs the bf16 inference
    engine.  ``dtype`` must be ``torch.bfloat16`` (fp16 is not implemented: bf16 is the native 16-bit type here)."""

    def __init__(self, enabled: bool = True, dtype: torch.dtype = torch.bfloat16, graph: Optional[bool] = None):
        """``graph``: run the VideoResNet trunk's bf16 forward and backward as two hipGraphs (captured once per clip shape: the
        trunk's ~560 launches per step become two graph launches; one forward per backward).  Default: the ``ZSV_AMP_GRAPH``
        environment variable (off)."""
        if enabled and dtype != torch.bfloat16:
            raise RuntimeError(f"amp.autocast: dtype {dtype} is not supported (bf16 only)")
        self.enabled = bool(enabled)
        self.graph = bool(int(os.environ.get("ZSV_AMP_GRAPH", "0") or 0)) if graph is None else bool(graph)

    def __enter__(self):
        self.prev = (getattr(_state, "enabled", False), getattr(_state, "graph", False))
        _state.enabled = self.enabled
        _state.graph = self.enabled and self.graph
        return self

    def __exit__(self, *exc):
        _state.enabled, _state.graph = self.prev
        return False


def is_autocast_enabled() -> bool:
    return getattr(_state, "enabled", False)


def is_graph_enabled() -> bool:
    return getattr(_state, "graph", False)


# ---- channels-last bf16 primitives -----------------------------------------------------------------------------------
def _rows(t: torch.Tensor) -> int:
    return t.numel() // t.shape[-1]


def conv_bf16_stats(d, x: torch.Tensor, blob: torch.Tensor):
    """The training forward of a convolution in front of a BatchNorm: ``(z, partials, rows)`` with the BatchNorm's batch statistics
    (per column tile: sum and sum of squares of the stored bf16 values, ``partials[:rows]`` of shape (rows, 2, Cp)) taken in the
    convolution's epilogue (``zsv_conv3d_bf16_fwd_stats``).  ``bn_cl_fwd_train(..., conv_stats=(partials, rows))`` consumes them."""
    import ctypes
    lib = _lib.load()
    cap = int(lib.zsv_conv3d_bf16_stat_rows(byref(d)))
    if cap <= 0:
        raise RuntimeError("zsv_conv3d_bf16_stat_rows: unsupported convolution geometry")
    cp = channel_pitch(d.Cout)
    z = torch.empty((d.N, d.To, d.Ho, d.Wo, cp), dtype=torch.bfloat16, device=x.device)
    partials = torch.empty((cap, 2, cp), dtype=torch.float32, device=x.device)
    rows = ctypes.c_int32(0)
    _lib.check(lib.zsv_conv3d_bf16_fwd_stats(byref(d), x.data_ptr(), blob.data_ptr(), z.data_ptr(), partials.data_ptr(), cap,
                                             ctypes.byref(rows), ops._stream()), "zsv_conv3d_bf16_fwd_stats")
    return z, partials, int(rows.value)


def bn_cl_fwd_train(z: torch.Tensor, bn: nn.BatchNorm3d, residual: Optional[torch.Tensor], relu: bool, want_coef: bool = False,
                    conv_stats=None):
    """Train-mode ``BatchNorm3d`` (+ residual) (+ ReLU) on a channels-last bf16 tensor.  Returns (y, mean, invstd), or
    (y, mean, invstd, coef) with ``want_coef``: the (2, Cp) scale / shift rows the backward recomputes the ReLU mask from.
    ``conv_stats`` = (partials, rows) of ``conv_bf16_stats``: the statistics pass over z is skipped."""
    lib = _lib.load()
    c = bn.num_features
    r = _rows(z)
    nbytes = lib.zsv_bn_cl_workspace_bytes(r, c)
    if nbytes == 0:
        raise RuntimeError(f"zsv_bn_cl: unsupported shape rows={r} channels={c}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=z.device)
    y = torch.empty_like(z)
    mean = torch.empty(c, dtype=torch.float32, device=z.device)
    invstd = torch.empty(c, dtype=torch.float32, device=z.device)
    coef = torch.empty((2, z.shape[-1]), dtype=torch.float32, device=z.device) if want_coef else None
    track = bn.track_running_stats and bn.running_mean is not None
    if bn.momentum is None and track:
        raise NotImplementedError("amp: BatchNorm3d(momentum=None) (cumulative moving average) is not used by the reference")
    momentum = 0.0 if bn.momentum is None else float(bn.momentum)
    if conv_stats is not None:
        partials, rows = conv_stats
        _lib.check(lib.zsv_bn_cl_fwd_train_stats(z.data_ptr(), ops._ptr(residual), r, c, ops._ptr(bn.weight), ops._ptr(bn.bias),
                                                 bn.running_mean.data_ptr() if track else None,
                                                 bn.running_var.data_ptr() if track else None, momentum, float(bn.eps),
                                                 1 if relu else 0, y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ops._ptr(coef),
                                                 partials.data_ptr(), rows, ws.data_ptr(), nbytes, ops._stream()),
                   "zsv_bn_cl_fwd_train_stats")
    else:
        _lib.check(lib.zsv_bn_cl_fwd_train(z.data_ptr(), ops._ptr(residual), r, c, ops._ptr(bn.weight), ops._ptr(bn.bias),
                                           bn.running_mean.data_ptr() if track else None,
                                           bn.running_var.data_ptr() if track else None, momentum, float(bn.eps),
                                           1 if relu else 0, y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ops._ptr(coef), ws.data_ptr(),
                                           nbytes, ops._stream()), "zsv_bn_cl_fwd_train")
    if track:
        pending = getattr(_state, "nbt_pending", None)
        if pending is not None:
            pending.append(bn.num_batches_tracked)       # one _foreach_add_ at the end of the trunk instead of 37 one-element launches
        else:
            bn.num_batches_tracked.add_(1)
    if want_coef:
        return y, mean, invstd, coef
    return y, mean, invstd


def bn_cl_bwd(dy: torch.Tensor, y: Optional[torch.Tensor], z: torch.Tensor, bn: nn.BatchNorm3d, mean, invstd, relu: bool,
              want_g: bool, fwd_coef: Optional[torch.Tensor] = None):
    """Returns (dz, g or None, dgamma, dbeta).  With ``fwd_coef`` (the forward's scale / shift rows; only valid when the forward
    had no residual) the ReLU mask is recomputed from z and the saved output ``y`` is not read."""
    lib = _lib.load()
    c = bn.num_features
    r = _rows(z)
    nbytes = lib.zsv_bn_cl_workspace_bytes(r, c)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=z.device)
    dz = torch.empty_like(z)
    g = torch.empty_like(z) if want_g else None
    dgamma = torch.empty(c, dtype=torch.float32, device=z.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=z.device)
    use_coef = relu and fwd_coef is not None
    _lib.check(lib.zsv_bn_cl_bwd(dy.data_ptr(), ops._ptr(y) if (relu and not use_coef) else None, z.data_ptr(), r, c, ops._ptr(bn.weight),
                                 mean.data_ptr(), invstd.data_ptr(), ops._ptr(fwd_coef) if use_coef else None, 1 if relu else 0,
                                 dz.data_ptr(), ops._ptr(g),
                                 dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), nbytes, ops._stream()), "zsv_bn_cl_bwd")
    return dz, g, dgamma, dbeta


def cl_to_ncdhw_f32(x: torch.Tensor, channels: int) -> torch.Tensor:
    """[N][T][H][W][Cp] bf16 -> (N, channels, T, H, W) fp32."""
    n, t, h, w, _ = x.shape
    out = torch.empty((n, channels, t, h, w), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().zsv_cl_bf16_to_ncs_f32(x.data_ptr(), n, t * h * w, channels, out.data_ptr(), ops._stream()),
               "zsv_cl_bf16_to_ncs_f32")
    return out


def ncdhw_to_cl_bf16(x: torch.Tensor) -> torch.Tensor:
    """(N, C, T, H, W) fp32 -> [N][T][H][W][Cp] bf16 (pad channels zero)."""
    ops._require(x)
    x = x.contiguous()
    n, c, t, h, w = x.shape
    out = torch.empty((n, t, h, w, channel_pitch(c)), dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.load().zsv_ncs_f32_to_cl_bf16(x.data_ptr(), n, t * h * w, c, out.data_ptr(), ops._stream()),
               "zsv_ncs_f32_to_cl_bf16")
    return out


def maxpool3d_bf16_bwd(dy: torch.Tensor, x: torch.Tensor, channels: int, kernel, padding) -> torch.Tensor:
    """Gradient of ``inference.maxpool3d_bf16`` (nn.MaxPool3d with stride == kernel) on channels-last bf16."""
    n, t, h, w, _ = x.shape
    kt, kh, kw = (int(v) for v in kernel)
    pt, ph, pw = (int(v) for v in padding)
    to, ho, wo = dy.shape[1:4]
    dx = torch.empty_like(x)
    _lib.check(_lib.load().zsv_maxpool3d_bf16_bwd(dy.data_ptr(), x.data_ptr(), n, channels, t, h, w, kt, kh, kw, pt, ph, pw, to, ho, wo,
                                                  dx.data_ptr(), ops._stream()), "zsv_maxpool3d_bf16_bwd")
    return dx


def relu_bias_bwd_cl(dy: torch.Tensor, y: torch.Tensor, channels: int, want_bias: bool = True):
    """``relu(conv(x) + bias)`` backward on channels-last bf16: (dy * (y > 0), bias gradient (fp32) or None)."""
    lib = _lib.load()
    r = _rows(y)
    nbytes = lib.zsv_bn_cl_workspace_bytes(r, channels)
    if nbytes == 0:
        raise RuntimeError(f"zsv_relu_bias_bwd_cl: unsupported shape rows={r} channels={channels}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=y.device)
    g = torch.empty_like(y)
    db = torch.empty(channels, dtype=torch.float32, device=y.device) if want_bias else None
    _lib.check(lib.zsv_relu_bias_bwd_cl(dy.data_ptr(), y.data_ptr(), r, channels, g.data_ptr(), ops._ptr(db), ws.data_ptr(), nbytes,
                                        ops._stream()), "zsv_relu_bias_bwd_cl")
    return g, db


def meanpool_bf16_bwd(dpooled: torch.Tensor, like: torch.Tensor, channels: int) -> torch.Tensor:
    n, t, h, w, _ = like.shape
    dx = torch.empty_like(like)
    _lib.check(_lib.load().zsv_meanpool_bf16_bwd(dpooled.contiguous().data_ptr(), n, t * h * w, channels, dx.data_ptr(),
                                                 ops._stream()), "zsv_meanpool_bf16_bwd")
    return dx


# ---- one Conv3d -> BatchNorm3d (-> + residual) (-> ReLU) unit -----------------------------------------------------------
class _Unit:
    def __init__(self, conv: nn.Conv3d, bn: Optional[nn.BatchNorm3d], relu: bool, plain: bool = False):
        """``plain``: C3D's ``relu(conv(x) + bias)`` (network.py:147-162): no BatchNorm, a bias."""
        if bn is None and not plain:
            raise RuntimeError("amp: every trunk convolution is followed by a BatchNorm3d in the reference's VideoResNet")
        if (conv.bias is not None and not plain) or tuple(conv.dilation) != (1, 1, 1) or conv.groups != 1:
            raise RuntimeError("amp: bias / dilation / groups are not used by the reference's trunks and not supported")
        self.conv, self.bn, self.relu = conv, bn, relu
        self.cout, self.cin = conv.weight.shape[0], conv.weight.shape[1]
        self.kernel = tuple(conv.weight.shape[2:])
        self.stride, self.padding = tuple(conv.stride), tuple(conv.padding)
        self.folded = self.cin <= 4            # the clip itself: border materialised, kw folded into K (conv_bf16.hip)

    def desc(self, n, t, h, w, folded_wo=None) -> ConvDesc:
        """``h, w``: stored extents of the input (with the materialised border for the clip convolution)."""
        kt, kh, kw = self.kernel
        pt, ph, pw = self.padding
        if self.folded:
            ph = pw = 0
        st, sh, sw = self.stride
        to = (t + 2 * pt - kt) // st + 1
        ho = (h + 2 * ph - kh) // sh + 1
        wo = (w + 2 * pw - kw) // sw + 1 if folded_wo is None else folded_wo
        return ConvDesc(n, self.cin, t, h, w, self.cout, to, ho, wo, kt, kh, kw, st, sh, sw, pt, ph, pw)


class _Record:
    __slots__ = ("unit", "x", "z", "y", "mean", "invstd", "desc", "has_res", "clips", "coef")


def _units(mods) -> List[_Unit]:
    return [_Unit(o._conv, o._bn, o.relu) for o in _conv_bn_relu_chain(list(mods), keep_modules=True)]


class Bf16TrainPath:
    """The trunk of a ``resnet.VideoResNet`` (BasicBlock models) as a list of Conv-BN units, with its forward tape and the
    backward walk.  Built once per trunk (``train_path_for``); holds no weights of its own (the fp32 parameters are packed to
    bf16 every step, after the optimizer moved them)."""

    def __init__(self, trunk: nn.Module):
        from . import resnet
        if not isinstance(trunk, resnet.VideoResNet):
            raise RuntimeError("amp: a resnet.VideoResNet trunk is expected")
        self.trunk = trunk
        self.stem = _units(trunk.stem)
        if not self.stem[0].folded:
            raise RuntimeError("amp: the stem's first convolution must take the clip (<= 4 channels)")
        self.blocks = []
        for layer in (trunk.layer1, trunk.layer2, trunk.layer3, trunk.layer4):
            for block in layer:
                if not isinstance(block, resnet.BasicBlock):
                    raise RuntimeError("amp: only BasicBlock trunks (the reference's *_18 models) are supported")
                main = _units(list(block.conv1)) + _units(list(block.conv2))
                main[-1].relu = True                 # out += residual; relu (resnet.py:110-111)
                down = _units(list(block.downsample)) if block.downsample is not None else None
                if down is not None and len(down) != 1:
                    raise RuntimeError("amp: the shortcut is one 1x1x1 convolution + BatchNorm (resnet.py:266-273)")
                self.blocks.append((main, down[0] if down else None))
        self.features = self.blocks[-1][0][-1].cout
        units = list(self.stem)
        for main, down in self.blocks:
            units += ([down] if down is not None else []) + main
        self.units = units
        # parameter order of the autograd Function: weight, gamma, beta per unit
        self.params = []
        for u in units:
            self.params += [p for p in (u.conv.weight, u.bn.weight, u.bn.bias) if p is not None]      # (affine=False: no gamma / beta)

    # -- forward -------------------------------------------------------------------------------------------------------
    def _unit_fwd(self, u: _Unit, x: torch.Tensor, tape, residual=None, clips=None, wo=None):
        n, t, h, w, _ = x.shape
        d = u.desc(n, t, h, w, wo)
        blob = pack_conv(d, u.conv.weight.detach(), None, None)
        timer = None if torch.cuda.is_current_stream_capturing() else ops.KERNEL_TIMER      # (timing events cannot be captured)
        mark = timer.start() if timer is not None and timer.wants("conv_bf16_fwd", d) else None
        # the BatchNorm's batch statistics come out of the convolution's epilogue (ZSV_AMP_NO_CONV_STATS=1: a pass over z instead)
        stats = None
        if os.environ.get("ZSV_AMP_NO_CONV_STATS"):
            z = conv_bf16(d, x, blob, None, False)
        else:
            z, partials, rows = conv_bf16_stats(d, x, blob)
            stats = (partials, rows)
        if mark is not None:
            timer.stop(mark)
        # (a unit with ReLU and no residual keeps its scale / shift rows: the backward recomputes the mask from z, y is not read there)
        keep_coef = tape is not None and u.relu and residual is None
        out = bn_cl_fwd_train(z, u.bn, residual, u.relu, want_coef=keep_coef, conv_stats=stats)
        y, mean, invstd = out[0], out[1], out[2]
        if tape is not None:
            r = _Record()
            r.unit, r.x, r.z, r.y, r.mean, r.invstd, r.desc, r.has_res, r.clips = u, x, z, y, mean, invstd, d, residual is not None, clips
            r.coef = out[3] if keep_coef else None
            tape.append(r)
        return y

    def forward(self, clips: torch.Tensor, tape):
        first = self.stem[0]
        n, _, t, h, w = clips.shape
        kt, kh, kw = first.kernel
        ph, pw = first.padding[1], first.padding[2]
        wo = (w + 2 * pw - kw) // first.stride[2] + 1
        hp, wp = h + 2 * ph, max(w + 2 * pw, (wo - 1) * first.stride[2] + 8)
        x = clip_to_bf16(clips, ph, pw, hp, wp)
        x = self._unit_fwd(first, x, tape, clips=clips, wo=wo)
        for u in self.stem[1:]:
            x = self._unit_fwd(u, x, tape)
        for main, down in self.blocks:
            residual = x if down is None else self._unit_fwd(down, x, tape)
            y = x
            for u in main[:-1]:
                y = self._unit_fwd(u, y, tape)
            x = self._unit_fwd(main[-1], y, tape, residual=residual)
        return x

    # -- backward pieces -----------------------------------------------------------------------------------------------
    @staticmethod
    def _axis_classes(k: int, p: int, s: int, n_in: int, n_out: int):
        """Residue classes of one axis of a convolution's input gradient.  Input position i = s*j + rho receives
        sum_m w[r + s*m] * dy[j + c - m] with r = (rho + p) mod s, c = (rho + p - r) / s: a stride-1 correlation of dy with the
        taps of that residue.  Per class: (rho, r, taps, symmetric padding for the forward kernel, first useful output, count).
        A stride-1 axis is the single class (0, 0, k, k-1-p, 0, n_in)."""
        out = []
        for rho in range(s):
            count = (n_in - rho + s - 1) // s
            if count <= 0:
                continue
            r = (rho + p) % s
            taps = (k - r + s - 1) // s if k > r else 0
            if taps == 0:
                out.append((rho, r, 0, 0, 0, count))
                continue
            c = (rho + p - r) // s
            pad = max(0, taps - 1 - c, c + count - n_out)
            out.append((rho, r, taps, pad, c - taps + 1 + pad, count))
        return out

    @staticmethod
    def _dgrad_problem(u: _Unit, dz: torch.Tensor, weight: torch.Tensor, kernel, pads, out_dims):
        """One stride-1 convolution of ``dz`` with the (channel-swapped, tap-flipped) ``weight`` on the forward kernel."""
        n, to, ho, wo, _ = dz.shape
        kt, kh, kw = kernel
        d2 = ConvDesc(n, u.cout, to, ho, wo, u.cin, out_dims[0], out_dims[1], out_dims[2], kt, kh, kw, 1, 1, 1, pads[0], pads[1], pads[2])
        lib = _lib.load()
        nbytes = lib.zsv_conv3d_bf16_blob_bytes(byref(d2))
        if nbytes == 0:
            raise RuntimeError("zsv_conv3d_bf16_blob_bytes: unsupported input-gradient geometry")
        blob = torch.empty(int(nbytes), dtype=torch.uint8, device=dz.device)
        # (channel roles swapped and taps flipped while packing: no transposed copy of the weight)
        _lib.check(lib.zsv_conv3d_bf16_pack_dgrad(byref(d2), weight.data_ptr(), blob.data_ptr(), ops._stream()),
                   "zsv_conv3d_bf16_pack_dgrad")
        return conv_bf16(d2, dz, blob, None, False)

    @staticmethod
    def _dgrad(r: _Record, dz: torch.Tensor) -> torch.Tensor:
        """Input gradient of the unit's convolution on the FORWARD kernel: a stride-1 convolution of the output gradient with the
        transposed, tap-flipped weights.  A strided convolution splits into one such problem per residue class of the input
        positions (stride (1,2,2): four classes with 1 / 2 / 2 / 4 of the nine taps on the compact output grid -- exactly the
        forward's multiply count, where a zero-interleaved output gradient would cost four times as much); the class results
        are interleaved into the input gradient."""
        u, d = r.unit, r.desc
        w = u.conv.weight.detach()
        axes = [Bf16TrainPath._axis_classes(k, p, s, n_in, n_out) for k, p, s, n_in, n_out in
                zip(u.kernel, u.padding, u.stride, (d.Ti, d.Hi, d.Wi), (d.To, d.Ho, d.Wo))]
        if all(len(a) == 1 for a in axes):                          # stride 1: the result is the gradient itself
            (ct, ch, cw) = (a[0] for a in axes)
            return Bf16TrainPath._dgrad_problem(u, dz, w.contiguous(), u.kernel, (ct[3], ch[3], cw[3]), (d.Ti, d.Hi, d.Wi))
        empty_class = any(c[2] == 0 for a in axes for c in a)
        alloc = torch.zeros if empty_class else torch.empty
        dx = alloc((d.N, d.Ti, d.Hi, d.Wi, channel_pitch(u.cin)), dtype=torch.bfloat16, device=dz.device)
        st, sh, sw = u.stride
        for ct in axes[0]:
            for ch in axes[1]:
                for cw in axes[2]:
                    if ct[2] == 0 or ch[2] == 0 or cw[2] == 0:
                        continue                                    # no tap reaches these positions (1x1x1 stride 2: 7 of 8 classes)
                    sub = w[:, :, ct[1]::st, ch[1]::sh, cw[1]::sw].contiguous()
                    dims = tuple(n_out + 2 * c[3] - c[2] + 1 for c, n_out in zip((ct, ch, cw), (d.To, d.Ho, d.Wo)))
                    part = Bf16TrainPath._dgrad_problem(u, dz, sub, (ct[2], ch[2], cw[2]), (ct[3], ch[3], cw[3]), dims)
                    dx[:, ct[0]::st, ch[0]::sh, cw[0]::sw] = part[:, ct[4]:ct[4] + ct[5], ch[4]:ch[4] + ch[5], cw[4]:cw[4] + cw[5]]
        return dx

    @staticmethod
    def _wgrad(r: _Record, dz: torch.Tensor) -> torch.Tensor:
        """fp32 accumulation of the bf16-rounded operands.  Stride-1 "same" convolutions (1x3x3, 3x1x1, 3x3x3: 90 % of the
        weight-gradient FLOPs) run ``zsv_conv3d_bf16_wgrad`` on the channels-last bf16 tensors as they are; the strided
        convolutions, the 1x1x1 shortcuts and the clip convolution convert the two operands to fp32 NCDHW
        (``zsv_cl_bf16_to_ncs_f32``) and use the fp32 weight-gradient kernels of the main path."""
        u, d = r.unit, r.desc
        lib = _lib.load()
        weight = u.conv.weight
        native = 0 if u.folded else int(lib.zsv_conv3d_bf16_wgrad_workspace_bytes(byref(d)))
        if native:
            x_cl = r.x

            def launch_native(stream):
                out = torch.empty_like(weight)
                ws = ops._workspace(native, dz.device)
                _lib.check(lib.zsv_conv3d_bf16_wgrad(byref(d), x_cl.data_ptr(), dz.data_ptr(), out.data_ptr(), ops._ptr(ws), native,
                                                     stream), "zsv_conv3d_bf16_wgrad")
                return out

            return ops._on_wgrad_stream(launch_native, (x_cl, dz), weight)
        if u.folded:
            x32 = r.clips.contiguous()                               # the fp32 clip itself (the stem has no bf16 copy in NCDHW)
        else:
            x32 = cl_to_ncdhw_f32(r.x, u.cin)
        df = ops.conv_desc(x32.shape, u.conv.weight.shape, u.stride, u.padding)
        dz32 = cl_to_ncdhw_f32(dz, u.cout)
        if (df.To, df.Ho, df.Wo) != tuple(dz32.shape[2:]):
            raise RuntimeError(f"amp: output gradient {tuple(dz32.shape)} does not match the convolution geometry")
        nbytes = lib.zsv_conv3d_wgrad_workspace_bytes(byref(df))

        def launch(stream):
            out = torch.empty_like(weight)
            ws = ops._workspace(nbytes, dz.device)
            _lib.check(lib.zsv_conv3d_wgrad(byref(df), x32.data_ptr(), dz32.data_ptr(), out.data_ptr(), ops._ptr(ws), nbytes,
                                            stream), "zsv_conv3d_wgrad")
            return out

        return ops._on_wgrad_stream(launch, (x32, dz32), weight)

    def backward(self, tape: List[_Record], dfeat: torch.Tensor, need_weight_grads=True):
        """``dfeat``: gradient of the last block's output (channels-last bf16).  Returns {parameter id: gradient}."""
        grads = {}
        idx = len(tape) - 1

        def unit_bwd(dy, want_g=False, need_dx=True):
            nonlocal idx
            r = tape[idx]
            idx -= 1
            u = r.unit
            dz, g, dgamma, dbeta = bn_cl_bwd(dy, r.y, r.z, u.bn, r.mean, r.invstd, u.relu, want_g, fwd_coef=r.coef)
            if u.bn.weight is not None:
                grads[id(u.bn.weight)] = dgamma
            if u.bn.bias is not None:
                grads[id(u.bn.bias)] = dbeta
            if need_weight_grads and u.conv.weight.requires_grad:
                grads[id(u.conv.weight)] = self._wgrad(r, dz)
            dx = self._dgrad(r, dz) if need_dx else None
            return dx, g, r

        dx = dfeat
        for main, down in reversed(self.blocks):
            # tail unit: BatchNorm + residual + ReLU -- the masked gradient is also the residual branch's gradient
            dy, g, r_tail = unit_bwd(dx, want_g=True)
            for _ in main[:-1]:
                dy, _, _ = unit_bwd(dy)
            if down is not None:
                dxd, _, _ = unit_bwd(g)
                dx = dy + dxd
            else:
                dx = dy + g
        for i in range(len(self.stem) - 1, -1, -1):
            dx, _, _ = unit_bwd(dx, need_dx=(i > 0))
        assert idx == -1
        return grads


class Bf16TrainPathC3D:
    """``network.C3D``'s eight ``relu(conv + bias)`` layers and five max-pools (network.py:147-163) forward and backward in bf16 on
    the channels-last layout; fc6 / dropout / clip mean / regressor / normalisation stay fp32 in the module (autocast's output
    of the last pool, flattened in the module's (C, T, H, W) order, is this path's result)."""

    def __init__(self, model: nn.Module):
        from .inference import Bf16EngineC3D
        self.model = model
        self.layers = []
        for conv_name, pool_name in Bf16EngineC3D.LAYERS:
            conv = getattr(model, conv_name)
            u = _Unit(conv, None, True, plain=True)
            pool = getattr(model, pool_name) if pool_name else None
            if pool is not None:
                k, st, pd = pool.kernel_size, pool.stride, pool.padding
                k = (k,) * 3 if isinstance(k, int) else tuple(k)
                st = k if st is None else ((st,) * 3 if isinstance(st, int) else tuple(st))
                pd = (pd,) * 3 if isinstance(pd, int) else tuple(pd)
                if st != k:
                    raise RuntimeError("amp: max-pools with stride != kernel are not used by the reference")
                pool = (k, pd)
            self.layers.append((u, pool))
        if not self.layers[0][0].folded:
            raise RuntimeError("amp: C3D's conv1 must take the clip (<= 4 channels)")
        self.features = self.layers[-1][0].cout
        self.params = []
        for u, _ in self.layers:
            self.params += [u.conv.weight, u.conv.bias]

    def forward(self, clips: torch.Tensor, tape):
        first = self.layers[0][0]
        n, _, t, h, w = clips.shape
        kt, kh, kw = first.kernel
        ph, pw = first.padding[1], first.padding[2]
        wo = (w + 2 * pw - kw) // first.stride[2] + 1
        hp, wp = h + 2 * ph, max(w + 2 * pw, (wo - 1) * first.stride[2] + 8)
        x = clip_to_bf16(clips, ph, pw, hp, wp)
        for i, (u, pool) in enumerate(self.layers):
            nn_, t_, h_, w_, _ = x.shape
            d = u.desc(nn_, t_, h_, w_, wo if i == 0 else None)
            bias = u.conv.bias.detach() if u.conv.bias is not None else None
            blob = pack_conv(d, u.conv.weight.detach(), None, bias)
            y = conv_bf16(d, x, blob, None, True)                    # relu(conv(x) + bias), one rounding to bf16
            r = _Record()
            r.unit, r.x, r.z, r.y, r.mean, r.invstd, r.desc, r.has_res, r.clips, r.coef = u, x, None, y, None, None, d, False, \
                (clips if i == 0 else None), None
            x = y
            pooled = None
            if pool is not None:
                pooled = maxpool3d_bf16_fwd(y, u.cout, pool[0], pool[1])
                x = pooled
            if tape is not None:
                tape.append((r, pool, pooled))
        return x

    def backward(self, tape, dlast: torch.Tensor):
        grads = {}
        dx = dlast
        for i in range(len(tape) - 1, -1, -1):
            r, pool, pooled = tape[i]
            u = r.unit
            if pool is not None:
                dx = maxpool3d_bf16_bwd(dx, r.y, u.cout, pool[0], pool[1])
            need_bias = u.conv.bias is not None and u.conv.bias.requires_grad
            g, db = relu_bias_bwd_cl(dx, r.y, u.cout, want_bias=need_bias)
            if need_bias:
                grads[id(u.conv.bias)] = db
            if u.conv.weight.requires_grad:
                grads[id(u.conv.weight)] = Bf16TrainPath._wgrad(r, g)
            dx = Bf16TrainPath._dgrad(r, g) if i > 0 else None
        return grads


def maxpool3d_bf16_fwd(x, channels, kernel, padding):
    from .inference import maxpool3d_bf16
    return maxpool3d_bf16(x, channels, kernel, padding)


class _C3DTrunkBf16(Function):
    """(N, 8192) fp32 = C3D's last pooled feature map flattened in (C, T, H, W) order (network.py:165), computed in bf16; fp32
    gradients for the convolution weights and biases, none for the clip."""

    @staticmethod
    def forward(ctx, clips, path, record, *params):
        ops._require(clips)
        tape = [] if record and any(p.requires_grad for p in params) else None
        with torch.cuda.device(clips.device):
            last = path.forward(clips.contiguous(), tape)
            n = last.shape[0]
            feat = last[..., :path.features].permute(0, 4, 1, 2, 3).reshape(n, -1).float()
        ctx.path, ctx.tape, ctx.last_shape = path, tape, tuple(last.shape)
        ctx.n_params = len(params)
        ctx.set_materialize_grads(False)
        return feat

    @staticmethod
    @once_differentiable
    def backward(ctx, dfeat):
        path, tape = ctx.path, ctx.tape
        if dfeat is None or tape is None:
            return (None, None, None) + (None,) * ctx.n_params
        n, t, h, w, _ = ctx.last_shape
        with torch.cuda.device(dfeat.device):
            dlast = ncdhw_to_cl_bf16(dfeat.float().reshape(n, path.features, t, h, w))
            grads = path.backward(tape, dlast)
        ctx.tape = None
        out = [grads.get(id(p)) if ctx.needs_input_grad[3 + i] else None for i, p in enumerate(path.params)]
        return (None, None, None) + tuple(out)


def c3d_features(model: nn.Module, clips: torch.Tensor) -> torch.Tensor:
    """``network.C3D.forward`` up to ``view(-1, 8192)`` (network.py:147-165) for (N, 3, 16, 112, 112) fp32 clips, in bf16."""
    if not clips.is_cuda:
        raise RuntimeError("amp: MI355X HIP tensors only (there is no CPU fallback)")
    path = model.__dict__.get("_zsv_bf16_train_path")
    if path is None:
        path = model.__dict__["_zsv_bf16_train_path"] = Bf16TrainPathC3D(model)
    return _C3DTrunkBf16.apply(clips, path, torch.is_grad_enabled(), *path.params)


class _GraphedTrunk:
    """The bf16 trunk's forward and backward as two hipGraphs (``torch.cuda.CUDAGraph``) for ONE clip shape: every launch of
    ``Bf16TrainPath.forward`` / ``.backward`` -- weight packs, convolutions, BatchNorm passes, the weight gradients forked to
    the side stream and joined at the end -- is captured once; a step replays the two graphs (12 us of host time each instead of
    ~8 ms of Python per pass, and no launch gaps on the device).  Inputs and outputs are static buffers: the clip is copied in,
    the pooled feature and the parameter gradients are copied out.  The kernels read the parameters, BatchNorm running
    statistics and counters through their (fixed) addresses, so the optimizer's updates are seen by the next replay.  One
    forward per backward: a second graphed forward before the first one's backward would overwrite its tape (checked)."""

    def __init__(self, path: "Bf16TrainPath", clips: torch.Tensor):
        self.path = path
        self.shape = tuple(clips.shape)
        dev = clips.device
        self.clips = clips.detach().clone()
        self.generation = 0
        bns = [u.bn for u in path.units]
        saved = [(bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()) for bn in bns if bn.running_mean is not None]

        def forward_body():
            tape = []
            _state.nbt_pending = []
            try:
                feat = path.forward(self.clips, tape)
            finally:
                pending, _state.nbt_pending = _state.nbt_pending, None
            if pending:
                torch._foreach_add_(pending, 1)
            return tape, feat, meanpool_bf16(feat, path.features)

        def backward_body(tape, feat, dpooled):
            dfeat = meanpool_bf16_bwd(dpooled, feat, path.features)
            with ops.deferred_wgrad_join():
                grads = path.backward(tape, dfeat)
            ops.join_wgrad_streams()
            return grads

        # warm-up on a side stream (one-time kernel attribute calls, allocator): it really runs, so the BatchNorm buffers are put back
        cur = torch.cuda.current_stream(dev)
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(cur)
        with torch.cuda.stream(warm):
            for _ in range(2):
                tape, feat, pooled = forward_body()
                backward_body(tape, feat, torch.zeros_like(pooled))
            del tape, feat, pooled
        cur.wait_stream(warm)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            i = 0
            for bn in bns:
                if bn.running_mean is not None:
                    bn.running_mean.copy_(saved[i][0]); bn.running_var.copy_(saved[i][1]); bn.num_batches_tracked.copy_(saved[i][2])
                    i += 1
        self.fwd_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph):
            self.tape, self.feat, self.pooled = forward_body()
        self.dpooled = torch.zeros_like(self.pooled)
        self.bwd_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.bwd_graph, pool=self.fwd_graph.pool()):
            self.grads = backward_body(self.tape, self.feat, self.dpooled)
        self.grad_list = [self.grads.get(id(p)) for p in path.params]

    def forward(self, clips: torch.Tensor) -> torch.Tensor:
        self.clips.copy_(clips)
        self.fwd_graph.replay()
        self.generation += 1
        return self.pooled.clone()

    def backward(self, dpooled: torch.Tensor, generation: int, needs):
        if generation != self.generation:
            raise RuntimeError("amp (graph mode): the graphed trunk ran forward again before this backward -- one forward per backward; "
                               "use amp.autocast(graph=False) for several forwards per backward pass")
        self.dpooled.copy_(dpooled)
        self.bwd_graph.replay()
        # the gradients live in the graph's static buffers: hand autograd copies (one flat buffer, one multi-tensor copy)
        src = [g for g, need in zip(self.grad_list, needs) if g is not None and need]
        flat = torch.empty(sum(g.numel() for g in src), dtype=torch.float32, device=dpooled.device)
        views, off = [], 0
        for g in src:
            views.append(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        if src:
            torch._foreach_copy_(views, src)
        it = iter(views)
        return [next(it) if (g is not None and need) else None for g, need in zip(self.grad_list, needs)]


def train_path_for(trunk: nn.Module) -> Bf16TrainPath:
    path = trunk.__dict__.get("_zsv_bf16_train_path")
    if path is None:
        path = trunk.__dict__["_zsv_bf16_train_path"] = Bf16TrainPath(trunk)
    return path


class _TrunkBf16(Function):
    """pooled (N, 512) fp32 = mean over voxels of the trunk's last feature map, computed in bf16; gradients for the clip are
    not produced (the clip is data), gradients for every trunk parameter are fp32."""

    @staticmethod
    def forward(ctx, clips, path, record, *params):
        ops._require(clips)
        ctx.graphed = None
        if record and is_graph_enabled() and all(p.requires_grad for p in params):
            graphs = path.__dict__.setdefault("_graphs", {})
            key = (tuple(clips.shape), clips.device.index)
            with torch.cuda.device(clips.device):
                g = graphs.get(key)
                if g is None:
                    while len(graphs) >= 2:                     # (a captured pair keeps its whole tape in a private pool: two shapes at most)
                        graphs.pop(next(iter(graphs)))
                    g = graphs[key] = _GraphedTrunk(path, clips.contiguous())
                pooled = g.forward(clips)
            ctx.graphed, ctx.generation, ctx.n_params = g, g.generation, len(params)
            ctx.set_materialize_grads(False)
            return pooled
        # (`record`: grad mode at the call site -- inside a Function's forward it is always off; no tape under torch.no_grad())
        tape = [] if record and any(p.requires_grad for p in params) else None
        with torch.cuda.device(clips.device):
            _state.nbt_pending = []
            try:
                feat = path.forward(clips.contiguous(), tape)
            finally:
                pending, _state.nbt_pending = _state.nbt_pending, None
            if pending:
                torch._foreach_add_(pending, 1)
            pooled = meanpool_bf16(feat, path.features)
        ctx.path, ctx.tape, ctx.feat_like = path, tape, feat
        ctx.n_params = len(params)
        ctx.set_materialize_grads(False)
        return pooled

    @staticmethod
    @once_differentiable
    def backward(ctx, dpooled):
        if ctx.graphed is not None:
            if dpooled is None:
                return (None, None, None) + (None,) * ctx.n_params
            with torch.cuda.device(dpooled.device):
                out = ctx.graphed.backward(dpooled.float(), ctx.generation, ctx.needs_input_grad[3:])
            return (None, None, None) + tuple(out)
        path, tape = ctx.path, ctx.tape
        if dpooled is None or tape is None:
            return (None, None, None) + (None,) * ctx.n_params
        with torch.cuda.device(dpooled.device):
            dfeat = meanpool_bf16_bwd(dpooled.float(), ctx.feat_like, path.features)
            grads = path.backward(tape, dfeat)
        ctx.tape = None
        out = []
        for i, p in enumerate(path.params):
            g = grads.get(id(p)) if ctx.needs_input_grad[3 + i] else None
            out.append(g)
        return (None, None, None) + tuple(out)


def trunk_features(trunk: nn.Module, clips: torch.Tensor) -> torch.Tensor:
    """``VideoResNet.forward``'s pooled output (resnet.py:251-254) for (N, 3, T, H, W) fp32 clips, trunk in bf16, train mode."""
    if not clips.is_cuda:
        raise RuntimeError("amp: MI355X HIP tensors only (there is no CPU fallback)")
    path = train_path_for(trunk)
    _lib.note_raw_write(parameters=False)               # BatchNorm running statistics are written through raw pointers
    return _TrunkBf16.apply(clips, path, torch.is_grad_enabled(), *path.params)
