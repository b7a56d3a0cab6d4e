"""CPU restatement of the reference's model path in plain PyTorch -- TEST INFRASTRUCTURE.

This is the oracle the HIP path is checked against and the ``cpu_baseline`` that
``bench.py`` times.  It restates, table-driven and independently written, what the
reference builds in ``resnet.py`` and ``network.py``:

* ``VideoTrunk``            <- ``resnet.VideoResNet``       (resnet.py:190-281)
* ``video_trunk(arch)``      <- ``r3d_18/mc3_18/r2plus1d_18`` (resnet.py:293-362)
* ``EmbeddingModel``         <- ``network.Model``            (network.py:472-600)
* ``C3DOracle``              <- ``network.C3D``              (network.py:95-180)
* ``oracle_network(opt)``    <- ``network.get_network``      (network.py:24-44)
* ``train_step`` / ``mse``   <- ``main.py:170-203`` step order with ``main_02.py:256``
  output handling

Module attribute names / Sequential indices are chosen so ``state_dict()`` keys are
identical to the reference's (checked by ``tests/test_oracle.py::test_restatement_is_pinned_to_the_imported_reference``
in the build container, and pinned for the GPU box by ``tests/golden``).  Arithmetic is
whatever ``torch.nn.functional`` does on CPU in the dtype of the parameters (fp32, or
fp64 after ``.double()``).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn
import torch.nn.functional as F

# per-architecture plan: stem spec, and per-stage conv kind
#   kind 'full' = 3x3x3, 'flat' = 1x3x3, 'split' = (1x3x3 -> BN -> ReLU -> 3x1x1)
_ARCH = {
    "r3d_18": dict(stem="plain", kinds=["full"] * 4),
    "mc3_18": dict(stem="plain", kinds=["full", "flat", "flat", "flat"]),
    "r2plus1d_18": dict(stem="split", kinds=["split"] * 4),
}
_STAGES = [(64, 1), (128, 2), (256, 2), (512, 2)]   # (planes, stride) resnet.py:217-220


def _conv(cin, cout, k, s, p) -> nn.Conv3d:
    return nn.Conv3d(cin, cout, kernel_size=k, stride=s, padding=p, bias=False)


def _unit(kind: str, cin: int, cout: int, mid: int, stride: int) -> nn.Module:
    """One 'conv builder' instance (resnet.py:18-76)."""
    if kind == "full":
        return _conv(cin, cout, (3, 3, 3), (stride,) * 3, (1, 1, 1))
    if kind == "flat":
        return _conv(cin, cout, (1, 3, 3), (1, stride, stride), (0, 1, 1))
    return nn.Sequential(                                   # resnet.py:37-53
        _conv(cin, mid, (1, 3, 3), (1, stride, stride), (0, 1, 1)),
        nn.BatchNorm3d(mid),
        nn.ReLU(inplace=True),
        _conv(mid, cout, (3, 1, 1), (stride, 1, 1), (1, 0, 0)),
    )


def _shortcut_stride(kind: str, stride: int) -> Tuple[int, int, int]:
    return (1, stride, stride) if kind == "flat" else (stride,) * 3   # resnet.py:32-34,55-57,74-76


class ResidualUnit(nn.Module):
    """resnet.BasicBlock (resnet.py:79-113)."""

    def __init__(self, kind: str, cin: int, planes: int, stride: int, shortcut: Optional[nn.Module]):
        super().__init__()
        mid = (cin * planes * 27) // (cin * 9 + 3 * planes)             # resnet.py:91
        self.conv1 = nn.Sequential(_unit(kind, cin, planes, mid, stride),
                                   nn.BatchNorm3d(planes), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(_unit(kind, planes, planes, mid, 1), nn.BatchNorm3d(planes))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = shortcut
        self.stride = stride

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.conv2(self.conv1(x))
        return self.relu(y + skip)


class VideoTrunk(nn.Module):
    """resnet.VideoResNet: returns ``(pooled (N,512), layer4 features)`` (resnet.py:243-256)."""

    def __init__(self, arch: str, num_classes: int = 400):
        super().__init__()
        plan = _ARCH[arch]
        if plan["stem"] == "split":                                      # resnet.py:179-187
            self.stem = nn.Sequential(
                _conv(3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3)), nn.BatchNorm3d(45), nn.ReLU(inplace=True),
                _conv(45, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)), nn.BatchNorm3d(64), nn.ReLU(inplace=True))
        else:                                                            # resnet.py:168-173
            self.stem = nn.Sequential(
                _conv(3, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3)), nn.BatchNorm3d(64), nn.ReLU(inplace=True))
        cin = 64
        for idx, ((planes, stride), kind) in enumerate(zip(_STAGES, plan["kinds"]), start=1):
            shortcut = None
            if stride != 1 or cin != planes:                             # resnet.py:268-273
                shortcut = nn.Sequential(_conv(cin, planes, 1, _shortcut_stride(kind, stride), 0),
                                         nn.BatchNorm3d(planes))
            units = [ResidualUnit(kind, cin, planes, stride, shortcut),
                     ResidualUnit(kind, planes, planes, 1, None)]
            setattr(self, f"layer{idx}", nn.Sequential(*units))
            cin = planes
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.fc = nn.Linear(512, num_classes)                            # built, never applied (resnet.py:254)
        for m in self.modules():                                         # resnet.py:226-236
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0.0, 0.01)
                nn.init.zeros_(m.bias)

    def stages(self, x) -> List[torch.Tensor]:
        outs = [self.stem(x)]
        for i in range(1, 5):
            outs.append(getattr(self, f"layer{i}")(outs[-1]))
        return outs

    def forward(self, x):
        f = self.stages(x)[-1]
        return self.avgpool(f).flatten(1), f


def video_trunk(arch: str):
    def factory(pretrained: bool = False, progress: bool = True, **kw):
        if pretrained:
            raise RuntimeError("offline: no pretrained weights (SURVEY F3)")
        return VideoTrunk(arch, **kw)
    return factory


class MLPHead(nn.Module):
    """network.MLP (network.py:603-618)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, last_activate=False):
        super().__init__()
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))
        self.num_layers = num_layers
        self.last_activate = last_activate

    def forward(self, x):
        for i, lin in enumerate(self.layers):
            x = lin(x)
            if i + 1 < self.num_layers:
                x = F.relu(x)
        return x


class EmbeddingModel(nn.Module):
    """network.Model: trunk -> mean over (T,H,W) -> MLP(512,512,300,2) -> L2 normalise."""

    def __init__(self, network, fixconvs=False, nopretrained=False):
        super().__init__()
        self.model = network(pretrained=nopretrained)                    # network.py:481
        if fixconvs:
            for p in self.model.parameters():
                p.requires_grad = False
        # parameters the reference constructs but never uses (network.py:500-517)
        self.d_model = 256
        self.num_sentences = 1
        self.t_pos_embeds = nn.Embedding(1, 512)
        self.special_tokens = nn.Embedding(1, 256)
        self.feature2input_proj = nn.Linear(512, 256)
        layer = nn.TransformerEncoderLayer(d_model=256, dim_feedforward=1024, nhead=8,
                                           dropout=0.1, activation="gelu")
        self.encoder = nn.TransformerEncoder(layer, num_layers=6, enable_nested_tensor=False)
        self.output2emb_proj = MLPHead(512, 512, 300, 2)
        nn.init.normal_(self.t_pos_embeds.weight)
        nn.init.xavier_uniform_(self.special_tokens.weight)

    def forward(self, x):
        bs, nc = x.shape[:2]
        _, feats = self.model(x.reshape(bs * nc, *x.shape[2:]))          # network.py:534-536
        pooled = feats.mean(dim=(2, 3, 4))                               # network.py:595
        return F.normalize(self.output2emb_proj(pooled)), None            # network.py:596,600


class C3DOracle(nn.Module):
    """network.C3D (network.py:95-180). ``nopretrained=True`` would read a pickle that is
    not available offline (the flag is always False in the reference's CLI)."""

    _CONVS = [("conv1", 3, 64), ("conv2", 64, 128), ("conv3a", 128, 256), ("conv3b", 256, 256),
              ("conv4a", 256, 512), ("conv4b", 512, 512), ("conv5a", 512, 512), ("conv5b", 512, 512)]

    def __init__(self, fixconvs=False, nopretrained=True):
        super().__init__()
        if nopretrained:
            raise RuntimeError("offline: ./assets/c3d.pickle is not available (network.py:129-130)")
        for name, cin, cout in self._CONVS:
            setattr(self, name, nn.Conv3d(cin, cout, kernel_size=3, padding=1))
            if name in ("conv1", "conv2", "conv3b", "conv4b", "conv5b"):
                idx = name[4]
                k = (1, 2, 2) if idx == "1" else (2, 2, 2)
                pad = (0, 1, 1) if idx == "5" else 0
                setattr(self, "pool" + idx, nn.MaxPool3d(kernel_size=k, stride=k, padding=pad))
        self.fc6 = nn.Linear(8192, 4096)
        self.fc7 = nn.Linear(4096, 4096)
        self.fc8 = nn.Linear(4096, 487)
        self.dropout = nn.Dropout(p=0.10)
        self.relu = nn.ReLU()
        self.softmax = nn.Softmax()
        self.regressor = nn.Linear(4096, 300)
        if fixconvs:
            for name in [c[0] for c in self._CONVS] + ["fc6"]:
                for p in getattr(self, name).parameters():
                    p.requires_grad = False

    def forward(self, x):
        bs, nc = x.shape[:2]
        h = x.reshape(bs * nc, *x.shape[2:])
        for name, _, _ in self._CONVS:
            h = F.relu(getattr(self, name)(h))
            if name in ("conv1", "conv2", "conv3b", "conv4b", "conv5b"):
                h = getattr(self, "pool" + name[4])(h)
        h = self.dropout(F.relu(self.fc6(h.reshape(-1, 8192))))
        h = h.reshape(bs, nc, -1).mean(1)
        return F.normalize(self.regressor(h), dim=-1)


def oracle_network(opt):
    """network.get_network (network.py:24-44): same substring dispatch, same error."""
    name = opt.network
    if "r3d" in name:
        trunk = video_trunk("r3d_18")
    elif "2plus1d" in name:
        trunk = video_trunk("r2plus1d_18")
    elif "c3d" in name:
        return C3DOracle(fixconvs=opt.fixconvs, nopretrained=opt.nopretrained)
    else:
        raise Exception("Network {} not available!".format(name))
    return EmbeddingModel(trunk, fixconvs=opt.fixconvs, nopretrained=opt.nopretrained)


def make_opt(network: str = "r2plus1d_18", fixconvs: bool = False, nopretrained: bool = False):
    return SimpleNamespace(network=network, fixconvs=fixconvs, nopretrained=nopretrained)


def embed(model: nn.Module, x: torch.Tensor) -> torch.Tensor:
    out = model(x)
    return out[0] if isinstance(out, tuple) else out


def train_step(model: nn.Module, optimizer: torch.optim.Optimizer, x: torch.Tensor,
               z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """zero_grad -> forward -> MSE -> backward -> step (main.py:170-203, without AMP)."""
    optimizer.zero_grad()
    y = embed(model, x)
    loss = F.mse_loss(y, z)
    loss.backward()
    optimizer.step()
    return y.detach(), loss.detach()


def compute_accuracy(pred: torch.Tensor, classes: torch.Tensor, true: torch.Tensor) -> Tuple[float, float]:
    """main.py:316-325 with scipy cosine cdist (top-1 / top-5 in percent)."""
    import numpy as np
    from scipy.spatial.distance import cdist
    order = cdist(pred.numpy(), classes.numpy(), "cosine").argsort(1)
    y = cdist(true.numpy(), classes.numpy(), "cosine").argmin(1)
    top1 = float(np.mean(order[:, 0] == y) * 100)
    top5 = float(np.mean([t in p for t, p in zip(y, order[:, :5])]) * 100)
    return top1, top5


def evaluate_protocol(pred, true, label, classes, splits: int = 10) -> Dict[str, float]:
    """What ``evaluate`` (main.py:224-313) computes once the embeddings are collected: accuracy over all
    classes (main.py:264-266) and, for ``opt.split == -1``, the ten seeded half-class splits
    (main.py:281-304): ``np.random.seed(split)``, the first half of a permutation of the class indices,
    samples whose label is in it, accuracy against THAT half of the class table; mean / std over the
    splits of top-1 and top-5.  numpy arrays in, python floats out."""
    import numpy as np
    pred, true, classes = (np.asarray(a, dtype=np.float32) for a in (pred, true, classes))
    label = np.asarray(label).astype(np.int64)
    acc, acc5 = compute_accuracy(torch.from_numpy(pred), torch.from_numpy(classes), torch.from_numpy(true))
    out = {"accuracy": acc, "accuracy_top5": acc5, "n": int(len(pred))}
    a1, a5 = np.zeros(splits), np.zeros(splits)
    for split in range(splits):
        np.random.seed(split)
        chosen = np.random.permutation(len(classes))[:len(classes) // 2]
        sel = np.isin(label, chosen)
        a1[split], a5[split] = compute_accuracy(torch.from_numpy(pred[sel]), torch.from_numpy(classes[chosen]),
                                                torch.from_numpy(true[sel]))
    if splits:
        out.update(split_accuracy=float(a1.mean()), split_accuracy_std=float(a1.std()),
                   split_accuracy_top5=float(a5.mean()), split_accuracy_top5_std=float(a5.std()))
    return out
