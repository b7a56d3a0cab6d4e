"""Synthetic clips, targets and a name-keyed deterministic weight initialiser.

No dataset or checkpoint can be fetched in the build/bench environment, so every
input of the hot path is generated here from fixed seeds:

* clips follow the reference's input contract ``(u8/255 - 1)/2`` in ``[-0.5, 0]``
  (reference ``auxiliary/transforms.py:116-117``; normalisation is commented out at
  ``transforms.py:47-52``) with layout ``(bs, n_clips, 3, T, H, W)``
  (``auxiliary/auxiliary_dataset.py:510``);
* targets are rows of a unit-norm class table, like the reference's L2-normalised mean
  Word2Vec vectors (``auxiliary/auxiliary_word2vec.py:28-32``);
* weights follow the reference's init rules (``resnet.py:226-236`` for the trunk,
  torch defaults elsewhere) but are drawn per tensor from a generator seeded by the
  ``state_dict`` key, so two different implementations of the same module tree
  (this package and the CPU oracle) can be given bit-identical parameters without
  shipping 147 MB of weights.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Mapping, Tuple

import torch

EMBED_DIM = 300


def _gen(seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    return g


def synthetic_clips(n: int, frames: int = 16, size: int = 112, n_clips: int = 1,
                    seed: int = 1234, rank: int = 0) -> torch.Tensor:
    """``(n, n_clips, 3, frames, size, size)`` float32 clips with u8 pixel statistics."""
    u8 = torch.randint(0, 256, (n, n_clips, 3, frames, size, size), dtype=torch.uint8,
                       generator=_gen(seed + rank))
    return (u8.to(torch.float32) / 255.0 - 1.0) / 2.0


def class_table(n_classes: int, seed: int = 4321) -> torch.Tensor:
    """Unit-norm ``(n_classes, 300)`` stand-in for the Word2Vec class embeddings."""
    e = torch.randn(n_classes, EMBED_DIM, generator=_gen(seed), dtype=torch.float32)
    return torch.nn.functional.normalize(e, dim=1)


def synthetic_targets(n: int, n_classes: int = 400, seed: int = 4321,
                      rank: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Labels ``(n,)`` int64 and their class embeddings ``(n, 300)``."""
    table = class_table(n_classes, seed)
    labels = torch.randint(0, n_classes, (n,), generator=_gen(seed + 7919 * (rank + 1)))
    return labels, table[labels].clone()


def synthetic_eval_set(n: int, n_classes: int, seed: int = 77, noise: float = 0.3,
                       broken: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """A stand-in for one test set of ``evaluate`` (main.py:224-313): class table ``(C, 300)``, labels
    ``(n,)``, true embeddings ``(n, 300)`` and "predicted" embeddings = the true one plus Gaussian noise
    (not normalised: the cosine distance must not depend on it), tuned so top-1 lands mid-range.  The
    first ``broken`` labels are -1 (failed loads, auxiliary_dataset.py:502-505)."""
    table = class_table(n_classes, seed)
    g = _gen(seed * 31 + n_classes)
    labels = torch.randint(0, n_classes, (n,), generator=g)
    true = table[labels].clone()
    pred = true * (0.5 + torch.rand(n, 1, generator=g)) + noise * torch.randn(n, EMBED_DIM, generator=g)
    if broken:
        labels[:broken] = -1
    return table, labels, true, pred


def _key_seed(key: str, seed: int) -> int:
    return (zlib.crc32(key.encode("utf-8")) << 8) ^ (seed * 0x9E3779B1)


def keyed_state_dict(template: Mapping[str, torch.Tensor], seed: int = 0,
                     bn_jitter: bool = False) -> Dict[str, torch.Tensor]:
    """New values for every entry of ``template`` (a ``state_dict``), keyed by name.

    Rules (reference file:line for what they reproduce):
      * 5-D weights under ``model.`` / ``stem`` / ``layerN`` (VideoResNet trunk):
        kaiming normal, fan_out, relu (``resnet.py:228``);
      * other 5-D weights (C3D convs, ``network.py:102-117``) and all 2-D Linear
        weights + their biases: torch default ``U(-1/sqrt(fan_in), 1/sqrt(fan_in))``,
        except the trunk's unused ``fc``: ``N(0, 0.01)``, bias 0 (``resnet.py:234-236``);
      * BatchNorm: weight 1, bias 0, running_mean 0, running_var 1
        (``resnet.py:231-233``); with ``bn_jitter`` they are perturbed so tests see
        non-trivial affine parameters and running statistics;
      * anything else (embeddings, LayerNorm of the dead Transformer encoder):
        ``N(0, 1)`` for >=2-D, ones/zeros for 1-D weight/bias.
    """
    out: Dict[str, torch.Tensor] = {}
    keys = set(template.keys())
    for key, ref in template.items():
        g = _gen(_key_seed(key, seed))
        shape = tuple(ref.shape)
        prefix, _, leaf = key.rpartition(".")
        is_bn = (prefix + ".running_mean") in keys
        in_trunk = (key.startswith("model.") or key.startswith("stem.")
                    or key.startswith("layer"))
        if leaf == "num_batches_tracked":
            val = torch.zeros(shape, dtype=ref.dtype)
        elif is_bn:
            if not bn_jitter:
                val = {"weight": torch.ones, "running_var": torch.ones}.get(leaf, torch.zeros)(shape)
            elif leaf == "weight":
                val = 0.5 + torch.rand(shape, generator=g)
            elif leaf == "running_var":
                val = 0.5 + torch.rand(shape, generator=g)
            else:
                val = 0.1 * torch.randn(shape, generator=g)
        elif ref.dim() == 5:
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            fan_out = shape[0] * shape[2] * shape[3] * shape[4]
            if in_trunk:
                val = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
            else:
                b = 1.0 / math.sqrt(fan_in)
                val = (torch.rand(shape, generator=g) * 2.0 - 1.0) * b
        elif ref.dim() == 2 and leaf == "weight" and (prefix + ".bias") in keys:
            if key.endswith("model.fc.weight") or key == "fc.weight":
                val = torch.randn(shape, generator=g) * 0.01
            else:
                b = 1.0 / math.sqrt(shape[1])
                val = (torch.rand(shape, generator=g) * 2.0 - 1.0) * b
        elif ref.dim() == 1 and leaf == "bias" and (prefix + ".weight") in keys \
                and template[prefix + ".weight"].dim() in (2, 5):
            w = template[prefix + ".weight"]
            if key.endswith("model.fc.bias") or key == "fc.bias":
                val = torch.zeros(shape)
            else:
                fan_in = int(w[0].numel())
                b = 1.0 / math.sqrt(fan_in)
                val = (torch.rand(shape, generator=g) * 2.0 - 1.0) * b
        elif ref.dim() >= 2:
            val = torch.randn(shape, generator=g)
            if "in_proj" in key or "linear" in key or "out_proj" in key:
                val = val * 0.02
        elif leaf == "weight":
            val = torch.ones(shape)
        else:
            val = torch.zeros(shape)
        out[key] = val.to(ref.dtype)
    return out
