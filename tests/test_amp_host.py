"""Host-side logic of the bf16 training path (``amp.py``) that needs no GPU: the residue-class decomposition of a strided
convolution's input gradient, checked against torch autograd in fp64 on the CPU (pure index arithmetic)."""
import itertools

import pytest
import torch
import torch.nn.functional as F

from zeroshotvideoclassification_amd.amp import Bf16TrainPath, autocast, is_autocast_enabled


@pytest.mark.parametrize("xs,cout,k,stride,pad", [
    ((2, 3, 4, 12, 12), 5, (1, 3, 3), (1, 2, 2), (0, 1, 1)),       # S2 / S5 / S8 (resnet.py:40-45 with stride 2)
    ((2, 3, 4, 6, 6), 5, (3, 1, 1), (2, 1, 1), (1, 0, 0)),         # T2 / T5 / T8
    ((2, 3, 4, 12, 12), 5, (1, 1, 1), (2, 2, 2), (0, 0, 0)),       # the strided 1x1x1 shortcut (resnet.py:270)
    ((1, 3, 4, 10, 10), 4, (3, 3, 3), (2, 2, 2), (1, 1, 1)),       # R3D-18 (resnet.py:23-30)
    ((1, 3, 3, 7, 9), 4, (1, 3, 3), (1, 2, 2), (0, 1, 1)),         # odd extents
    ((1, 3, 5, 7, 9), 4, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ((1, 3, 5, 7, 9), 4, (3, 3, 3), (1, 1, 1), (1, 1, 1)),         # stride 1: one class, the ordinary flipped convolution
    ((1, 2, 4, 11, 11), 4, (3, 7, 7), (1, 2, 2), (1, 3, 3)),       # the stems' geometry
    ((1, 2, 4, 11, 12), 4, (1, 5, 4), (1, 3, 2), (0, 2, 1)),       # stride 3, even kernel: the algebra is general
])
def test_input_gradient_residue_classes_reproduce_autograd(xs, cout, k, stride, pad):
    n, cin, t, h, w = xs
    g = torch.Generator().manual_seed(sum(xs) + cout)
    wt = torch.randn((cout, cin) + tuple(k), dtype=torch.float64, generator=g)
    x = torch.zeros(xs, dtype=torch.float64, requires_grad=True)
    y = F.conv3d(x, wt, None, stride, pad)
    dz = torch.randn(y.shape, dtype=torch.float64, generator=g)
    y.backward(dz)
    out_dims = tuple(y.shape[2:])
    axes = [Bf16TrainPath._axis_classes(kk, p, s, ni, no) for kk, p, s, ni, no in zip(k, pad, stride, (t, h, w), out_dims)]
    dx = torch.zeros(xs, dtype=torch.float64)
    covered = torch.zeros(xs, dtype=torch.int32)
    for ct, ch, cw in itertools.product(*axes):
        covered[:, :, ct[0]::stride[0], ch[0]::stride[1], cw[0]::stride[2]] += 1
        if 0 in (ct[2], ch[2], cw[2]):
            continue                                                  # no tap reaches these positions: zero gradient
        sub = wt[:, :, ct[1]::stride[0], ch[1]::stride[1], cw[1]::stride[2]]
        assert tuple(sub.shape[2:]) == (ct[2], ch[2], cw[2])
        # what zsv_conv3d_bf16_pack_dgrad + the forward kernel compute: channel roles swapped, taps flipped, stride 1
        part = F.conv3d(dz, sub.transpose(0, 1).flip(2, 3, 4), None, 1, (ct[3], ch[3], cw[3]))
        assert tuple(part.shape[2:]) == tuple(no + 2 * c[3] - c[2] + 1 for c, no in zip((ct, ch, cw), out_dims))
        dx[:, :, ct[0]::stride[0], ch[0]::stride[1], cw[0]::stride[2]] = \
            part[:, :, ct[4]:ct[4] + ct[5], ch[4]:ch[4] + ch[5], cw[4]:cw[4] + cw[5]]
    assert int(covered.min()) == 1 and int(covered.max()) == 1        # the classes partition the input positions
    assert float((dx - x.grad).abs().max()) < 1e-10


def test_autocast_flag_nests_and_refuses_fp16():
    assert not is_autocast_enabled()
    with autocast():
        assert is_autocast_enabled()
        with autocast(enabled=False):
            assert not is_autocast_enabled()
        assert is_autocast_enabled()
    assert not is_autocast_enabled()
    with pytest.raises(RuntimeError):
        autocast(dtype=torch.float16)
