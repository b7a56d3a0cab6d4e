"""Clip pre-processing on the GPU behind the reference's ``get_transform`` surface.

``get_transform(is_validation, crop_size=112)`` (auxiliary/transforms.py:41-56) returns a callable
like the reference's ``Compose``; the difference is what it accepts and where it runs: the
reference transforms one ``(T, H, W, 3)`` uint8 clip on a CPU worker and ships fp32 to the GPU
(53 MB per 22-clip batch); here a batch of uint8 clips (13 MB) is uploaded and the whole chain
-- (u8/255-1)/2, THWC->CTHW, bilinear short-side-128 resize, 112 crop, horizontal flip -- is one
HIP kernel writing the ``(N, 3, T, 112, 112)`` model input.  Random crop / flip parameters are
drawn on the host with Python's ``random`` exactly like ``RandomCrop.get_params`` /
``RandomHorizontalFlip`` (transforms.py:137-147,192-195), one draw per clip.
"""
from __future__ import annotations

import random
from ctypes import c_void_p
from typing import Optional, Sequence

import torch

from . import _lib


def resized_hw(h: int, w: int, size: int):
    """Output size and source step of ``resize(vid, size)`` (transforms.py:99-107): scale =
    size / min(h, w) handed to ``F.interpolate(scale_factor=...)``."""
    scale = float(size) / min(h, w)
    return int(h * scale), int(w * scale), 1.0 / scale       # floor(in * scale); torch keeps 1/scale as float


class ClipTransform:
    def __init__(self, is_validation: bool, crop_size: int = 112):
        self.is_validation = bool(is_validation)
        self.crop_size = int(crop_size)
        self.size = 128 if crop_size == 112 else 256           # transforms.py:42

    def draw_params(self, n: int, hres: int, wres: int):
        th = tw = self.crop_size
        rows = []
        for _ in range(n):
            if self.is_validation:                               # CenterCrop (transforms.py:80-85)
                i, j = int(round((hres - th) / 2.)), int(round((wres - tw) / 2.))
                flip = 0
            else:                                                # RandomCrop + RandomHorizontalFlip
                i = 0 if hres == th and wres == tw else random.randint(0, hres - th)
                j = 0 if hres == th and wres == tw else random.randint(0, wres - tw)
                flip = 1 if random.random() < 0.5 else 0
            rows.append((i, j, flip))
        return rows

    def __call__(self, frames_u8: torch.Tensor, params: Optional[Sequence[Sequence[int]]] = None) -> torch.Tensor:
        """``frames_u8``: ``(N, T, H, W, 3)`` (or one clip ``(T, H, W, 3)``) uint8 on a HIP device.
        Returns ``(N, 3, T, crop, crop)`` fp32 (``(3, T, crop, crop)`` for a single clip)."""
        single = frames_u8.dim() == 4
        if single:
            frames_u8 = frames_u8.unsqueeze(0)
        if frames_u8.dim() != 5 or frames_u8.shape[-1] != 3 or frames_u8.dtype != torch.uint8:
            raise RuntimeError("expected (N, T, H, W, 3) uint8 frames")
        if not frames_u8.is_cuda:
            raise RuntimeError("ClipTransform runs on an MI355X HIP device only (no CPU fallback; "
                               "the CPU restatement lives in oracle/)")
        frames_u8 = frames_u8.contiguous()
        n, t, h, w, _ = (int(v) for v in frames_u8.shape)
        hres, wres, inv_scale = resized_hw(h, w, self.size)
        if hres < self.crop_size or wres < self.crop_size:
            raise RuntimeError("clip too small for the crop")
        if params is None:
            params = self.draw_params(n, hres, wres)
        if len(params) != n:
            raise RuntimeError("one (top, left, flip) triple per clip expected")
        for (i, j, _f) in params:
            if not (0 <= i <= hres - self.crop_size and 0 <= j <= wres - self.crop_size):
                raise RuntimeError("crop window outside the resized frame")
        # pinned + non_blocking: a pageable upload would drain the stream every call (the host could no longer run ahead)
        ptab = torch.tensor(params, dtype=torch.int32).pin_memory().to(frames_u8.device, non_blocking=True)
        out = torch.empty((n, 3, t, self.crop_size, self.crop_size), dtype=torch.float32, device=frames_u8.device)
        with torch.cuda.device(frames_u8.device):
            _lib.check(_lib.load().zsv_clip_transform(frames_u8.data_ptr(), n, t, h, w, hres, wres, float(inv_scale),
                                                      self.crop_size, ptab.data_ptr(), out.data_ptr(),
                                                      c_void_p(torch.cuda.current_stream().cuda_stream)),
                       "zsv_clip_transform")
        return out[0] if single else out


def get_transform(is_validation, crop_size=112):
    return ClipTransform(is_validation, crop_size)
