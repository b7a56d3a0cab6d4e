"""CPU restatement of the reference's clip transform chain -- TEST INFRASTRUCTURE.

Restates ``auxiliary/transforms.py`` with the same torch calls the reference makes:
``to_normalized_float_tensor`` (:116-117), ``resize`` (:99-107, ``F.interpolate`` bilinear,
align_corners=False, scale_factor = size / short side), ``crop`` (:60-61), ``center_crop``
(:80-85), ``hflip`` (:88-89) and ``get_transform``'s order (:41-56: normalise -> resize -> crop
-> flip).  Crop / flip parameters are explicit so the GPU path can be driven with the same ones.
Pinned to the reference: ``oracle/make_golden.py::run_transforms`` imports the reference's file in the build
container (``oracle/reference_import.import_reference_transforms``: ``imageio`` and
``torchvision.transforms.Compose`` are absent offline and carry no pixel arithmetic, so an empty module and an
apply-in-order ``Compose`` stand in), asserts that this restatement equals ``get_transform(True/False)`` bit for
bit and stores the reference's outputs in ``tests/golden/transforms.npz``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def to_normalized_float_tensor(vid_u8: torch.Tensor) -> torch.Tensor:
    return (vid_u8.permute(3, 0, 1, 2).to(torch.float32) / 255 - 1.0) / 2.0


def resize(vid: torch.Tensor, size: int) -> torch.Tensor:
    scale = float(size) / min(vid.shape[-2:])
    return F.interpolate(vid, size=None, scale_factor=scale, mode="bilinear", align_corners=False)


def center_crop_params(h: int, w: int, th: int, tw: int):
    return int(round((h - th) / 2.)), int(round((w - tw) / 2.))


def clip_transform(vid_u8: torch.Tensor, top: int, left: int, flip: bool, crop: int = 112, size: int = 128) -> torch.Tensor:
    """One ``(T, H, W, 3)`` uint8 clip -> ``(3, T, crop, crop)`` fp32."""
    v = resize(to_normalized_float_tensor(vid_u8), size)
    v = v[..., top:top + crop, left:left + crop]
    return v.flip(dims=(-1,)) if flip else v
