"""Fused multi-tensor Adam: ``torch.optim.Adam`` semantics (main.py:131), one kernel launch.

The reference steps ``torch.optim.Adam(model.parameters(), lr)`` (betas (0.9, 0.999), eps 1e-8,
no weight decay, no amsgrad).  ``FusedAdam`` keeps that update rule and the same ``state_dict``
layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter) but applies it to every parameter
that has a gradient with ONE launch of ``zsv_adam_multi``: a descriptor
table {p, g, m, v, n, first_chunk} is built on the host per step (gradients are re-allocated by
autograd every step, so their addresses change), uploaded, and a grid of
4096-element chunks walks all tensors.  Parameters without a gradient (the reference's dead
Transformer encoder etc.) are skipped exactly like torch does.
"""
from __future__ import annotations

import struct
from ctypes import c_void_p

import torch

from . import _lib

_CHUNK = 4096


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._ring = [[None, None] for _ in range(4)]     # (pinned staging buffer, copy-done event)
        self._next = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            entries = []
            first = 0
            step_no = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous fp32 parameters on a HIP device (no CPU fallback)")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                s = int(st["step"].item())
                if step_no is None:
                    step_no = s
                elif s != step_no:
                    raise RuntimeError("FusedAdam: parameters of one group must share the step count")
                n = p.numel()
                entries.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), n, first, g))
                first += (n + _CHUNK - 1) // _CHUNK
            if not entries:
                continue
            raw = bytearray(b"".join(struct.pack("<QQQQqq", *e[:6]) for e in entries))
            dev = entries[0][6].device
            # The host runs ahead of the GPU, so a pinned staging buffer may not be rewritten until the
            # copy queued from it has executed: rotate over a small ring guarded by events (a pageable
            # copy would be safe too, but torch synchronises the stream for it and the run-ahead is lost).
            nbytes = len(raw)
            slot = self._ring[self._next % len(self._ring)]
            self._next += 1
            if slot[0] is None or slot[0].numel() < nbytes:
                slot[0] = torch.empty(max(nbytes, 48 * 512), dtype=torch.uint8).pin_memory()
            if slot[1] is not None:
                slot[1].synchronize()
            slot[0][:nbytes].copy_(torch.frombuffer(raw, dtype=torch.uint8))
            table = slot[0][:nbytes].to(dev, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(torch.cuda.current_stream(dev))
            with torch.cuda.device(dev):
                _lib.check(lib.zsv_adam_multi(table.data_ptr(), len(entries), first, float(group["lr"]),
                                              float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                                              step_no, c_void_p(torch.cuda.current_stream().cuda_stream)),
                           "zsv_adam_multi")
            # keep the uploaded table and any contiguous gradient copies alive until the stream is past the launch
            table.record_stream(torch.cuda.current_stream())
            _lib.note_raw_write()                  # parameters updated through raw pointers
        return loss
