"""MI355X-native model wrappers behind the reference's ``network.py`` surface.

Drop-in for ``network.get_network(opt)`` and the modules it returns (network.py:24-44):
``model(X)`` at main.py:174,250 / main_02.py:256,436 keeps working unchanged -- same
constructor signatures, same ``state_dict`` keys (including the parameters the reference
builds but never uses), same outputs:

* ``Model.forward(x (bs,nc,3,T,H,W)) -> (emb (bs*nc,300), None)``   network.py:533-600
* ``C3D.forward(x) -> emb (bs,300)``                                network.py:143-180
* ``ResNet18`` (dead in the reference, kept for the surface)        network.py:50-80
* ``MLP``                                                           network.py:603-618

Convolutions, BatchNorm, ReLU, pooling and the dense layers run in the gfx950 kernels
(``ops``); ``F.normalize``, dropout and the clip-mean stay in PyTorch, as does autograd.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from . import resnet as models
from .layers import Conv3d, Linear, MaxPool3d, ReLU


def get_network(opt):
    """Selection function for available networks (network.py:24-44): substring dispatch on
    ``opt.network``; reads ``opt.fixconvs`` and ``opt.nopretrained``."""
    name = opt.network
    if "r3d" in name:
        factory = models.r3d_18
    elif "2plus1d" in name:
        factory = models.r2plus1d_18
    elif "c3d" in name:
        return C3D(fixconvs=opt.fixconvs, nopretrained=opt.nopretrained)
    else:
        raise Exception("Network {} not available!".format(name))
    return Model(factory, fixconvs=opt.fixconvs, nopretrained=opt.nopretrained)


def _freeze(module: nn.Module) -> None:
    for p in module.parameters():
        p.requires_grad = False


class MLP(nn.Module):
    """Linear(+ReLU) x (num_layers-1) + Linear (network.py:603-618); the ReLU is fused into the
    GEMM epilogue."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, last_activate=False):
        super().__init__()
        self.num_layers = num_layers
        self.last_activate = last_activate
        widths = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(Linear(a, b) for a, b in zip(widths[:-1], widths[1:]))

    def forward(self, x):
        last = self.num_layers - 1
        for i, layer in enumerate(self.layers):
            x = layer(x, relu=(i < last))
        return x


class Model(nn.Module):
    """Video trunk -> mean over (T,H,W) -> MLP(512,512,300,2) -> L2 normalise.
    Returns ``(embeddings, None)`` like the fork (network.py:600)."""

    def __init__(self, network, fixconvs=False, nopretrained=False):
        super().__init__()
        self.model = network(pretrained=nopretrained)        # network.py:481 (flag is always False)
        if fixconvs:
            _freeze(self.model)
        # Built by the reference and never used in forward (network.py:500-517): they exist so that
        # checkpoints round-trip key-for-key; they never receive gradients.
        self.d_model = 256
        self.num_sentences = 1
        self.t_pos_embeds = nn.Embedding(self.num_sentences, 512)
        self.special_tokens = nn.Embedding(1, self.d_model)
        self.feature2input_proj = nn.Linear(512, self.d_model)
        self.encoder = nn.TransformerEncoder(
            nn.TransformerEncoderLayer(d_model=self.d_model, dim_feedforward=self.d_model * 4, nhead=8, dropout=0.1,
                                       activation="gelu"),
            num_layers=6, enable_nested_tensor=False)
        self.output2emb_proj = MLP(512, 512, 300, 2)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.t_pos_embeds.weight)
        nn.init.xavier_uniform_(self.special_tokens.weight)

    def forward(self, x):
        bs, nc, ch, t, h, w = x.shape
        from . import amp
        if amp.is_autocast_enabled() and x.is_cuda:            # main.py:172 `with autocast():` -> the bf16 paths
            from . import resnet
            if isinstance(self.model, resnet.VideoResNet):
                if self.training:
                    pooled = amp.trunk_features(self.model, x.reshape(bs * nc, ch, t, h, w))
                    return F.normalize(self.output2emb_proj(pooled)), None
                if not torch.is_grad_enabled():
                    from .inference import engine_for
                    return engine_for(self, torch.bfloat16)(x)
        _, feats = self.model(x.reshape(bs * nc, ch, t, h, w))
        pooled = ops.mean_pool(feats)                          # torch.mean(feats, dim=(2,3,4)), network.py:595
        emb = F.normalize(self.output2emb_proj(pooled))        # network.py:596
        return emb, None


class ResNet18(nn.Module):
    """The original single-Linear head (network.py:50-80); unreachable from ``get_network`` in the
    fork but part of the module surface."""

    def __init__(self, network, fixconvs=False, nopretrained=True):
        super().__init__()
        self.model = network(pretrained=nopretrained)
        if fixconvs:
            _freeze(self.model)
        self.regressor = Linear(self.model.fc.in_features, 300)
        self.dropout = nn.Dropout(p=0.05)

    def forward(self, x):
        bs, nc, ch, t, h, w = x.shape
        pooled, _ = self.model(x.reshape(bs * nc, ch, t, h, w))
        pooled = pooled.reshape(bs, nc, -1).mean(1)
        return F.normalize(self.regressor(self.dropout(pooled)))


class C3D(nn.Module):
    """C3D (network.py:95-180): 8 x (Conv3d 3x3x3 + bias + ReLU, fused), 5 max-pools, fc6 + ReLU,
    dropout, clip mean, regressor, L2 normalise.  Returns a tensor, not a tuple."""

    def __init__(self, fixconvs=False, nopretrained=True):
        super().__init__()
        three = (3, 3, 3)
        one = (1, 1, 1)
        self.conv1 = Conv3d(3, 64, kernel_size=three, padding=one)
        self.pool1 = MaxPool3d(kernel_size=(1, 2, 2), stride=(1, 2, 2))
        self.conv2 = Conv3d(64, 128, kernel_size=three, padding=one)
        self.pool2 = MaxPool3d(kernel_size=(2, 2, 2), stride=(2, 2, 2))
        self.conv3a = Conv3d(128, 256, kernel_size=three, padding=one)
        self.conv3b = Conv3d(256, 256, kernel_size=three, padding=one)
        self.pool3 = MaxPool3d(kernel_size=(2, 2, 2), stride=(2, 2, 2))
        self.conv4a = Conv3d(256, 512, kernel_size=three, padding=one)
        self.conv4b = Conv3d(512, 512, kernel_size=three, padding=one)
        self.pool4 = MaxPool3d(kernel_size=(2, 2, 2), stride=(2, 2, 2))
        self.conv5a = Conv3d(512, 512, kernel_size=three, padding=one)
        self.conv5b = Conv3d(512, 512, kernel_size=three, padding=one)
        self.pool5 = MaxPool3d(kernel_size=(2, 2, 2), stride=(2, 2, 2), padding=(0, 1, 1))
        self.fc6 = Linear(8192, 4096)
        self.fc7 = Linear(4096, 4096)       # unused by forward (network.py:168-172)
        self.fc8 = Linear(4096, 487)        # unused by forward
        self.dropout = nn.Dropout(p=0.10)
        self.relu = ReLU()
        self.softmax = nn.Softmax()
        if nopretrained:                     # network.py:129-130 (sic: the flag name is inverted upstream)
            self.load_state_dict(torch.load("./assets/c3d.pickle"))
        self.regressor = Linear(4096, 300)
        if fixconvs:
            for m in (self.conv1, self.conv2, self.conv3a, self.conv3b, self.conv4a, self.conv4b, self.conv5a,
                      self.conv5b, self.fc6):
                _freeze(m)

    def forward(self, x):
        bs, nc, ch, t, h, w = x.shape
        from . import amp
        if amp.is_autocast_enabled() and x.is_cuda:            # main.py:172 `with autocast():`
            if not torch.is_grad_enabled() and not self.training:
                from .inference import engine_for
                return engine_for(self, torch.bfloat16)(x)      # eval forward: the bf16 engine
            if torch.is_grad_enabled():
                # the mixed-precision step: convolutions + pools forward and backward in bf16 (amp.Bf16TrainPathC3D), the head in fp32
                a = amp.c3d_features(self, x.reshape(bs * nc, ch, t, h, w))
                a = self.fc6(a, relu=True)
                a = self.dropout(a)
                a = a.reshape(bs, nc, -1).mean(1).reshape(bs, -1)
                return F.normalize(self.regressor(a), dim=-1)
        a = x.reshape(bs * nc, ch, t, h, w)
        a = self.pool1(self.conv1(a, relu=True))
        a = self.pool2(self.conv2(a, relu=True))
        a = self.pool3(self.conv3b(self.conv3a(a, relu=True), relu=True))
        a = self.pool4(self.conv4b(self.conv4a(a, relu=True), relu=True))
        a = self.pool5(self.conv5b(self.conv5a(a, relu=True), relu=True))
        a = self.fc6(a.reshape(-1, 8192), relu=True)
        a = self.dropout(a)
        a = a.reshape(bs, nc, -1).mean(1).reshape(bs, -1)
        return F.normalize(self.regressor(a), dim=-1)
