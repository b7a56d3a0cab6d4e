"""Fused multi-tensor Adam and an on-device loss scaler (main.py:131-137,195-203).

The reference steps ``torch.optim.Adam(model.parameters(), lr)`` (betas (0.9, 0.999), eps 1e-8, no weight
decay, no amsgrad) under ``torch.cuda.amp.GradScaler``: ``scaler.scale(loss).backward();
scaler.step(optimizer); scaler.update()``.

``FusedAdam`` keeps that update rule and the same ``state_dict`` layout (``step`` / ``exp_avg`` /
``exp_avg_sq`` per parameter) but applies it to every parameter that has a gradient with ONE launch of
``zsv_adam_multi`` driven by a descriptor table {p, g, m, v, n, first_chunk}:

* default: the table is rebuilt on the host every step (autograd re-allocates the gradients, so their addresses
  change), uploaded through a small pinned ring, and a grid of 4096-element chunks walks all tensors;
* ``grad_buckets=<ddp.GradientSync>``: the gradients live in the flat all-reduce buckets of the data-parallel
  exchange (``.grad`` are views of them), ``exp_avg`` / ``exp_avg_sq`` are allocated as matching flat buffers
  (the per-parameter state entries are views), and the table is built and uploaded ONCE per bucket layout --
  Adam reads the all-reduce buffers directly, no pack / unpack copies and no per-step host work.

Parameters without a gradient (the reference's dead Transformer encoder etc.) are skipped exactly
like torch does.

``LossScaler`` is ``GradScaler`` with its state (scale, growth tracker, found-inf flag, count of steps
actually taken) in device memory: the non-finite check (``zsv_grad_check_multi``), the skip-on-inf Adam step
with the 1/scale folded into the gradient read (``zsv_adam_multi_scaled``) and the scale update
(``zsv_scaler_update``) are three launches and no host synchronisation.
"""
from __future__ import annotations

import struct
from ctypes import c_void_p
from typing import Optional

import torch

from . import _lib

_CHUNK = 4096


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


class LossScaler:
    """``torch.cuda.amp.GradScaler`` (main.py:137) with device-resident state; same defaults."""

    def __init__(self, init_scale: float = 2.0 ** 16, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("LossScaler keeps its state on an MI355X HIP device (no CPU fallback)")
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # zsv_scaler_state {float scale; int32 growth_tracker; int32 found_inf; int32 steps_done}
        host = torch.zeros(4, dtype=torch.int32)
        host.view(torch.float32)[0] = float(init_scale)
        self._state = host.to(self.device)
        self._scale = self._state.view(torch.float32)[0:1]

    @property
    def state_ptr(self) -> int:
        return self._state.data_ptr()

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        """``scaler.scale(loss)`` (main.py:195): multiplied on the device, differentiable."""
        return loss * self._scale.detach().reshape(())

    def step(self, optimizer) -> None:
        """``scaler.step(optimizer)`` (main.py:200): unscale + non-finite check + (skipped-if-inf) Adam step."""
        if not isinstance(optimizer, FusedAdam):
            raise RuntimeError("LossScaler.step drives optim.FusedAdam (the check, unscale and skip run inside its kernels)")
        optimizer.step(scaler=self)

    def update(self) -> None:
        """``scaler.update()`` (main.py:203)."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().zsv_scaler_update(self.state_ptr, self.growth_factor, self.backoff_factor,
                                                     self.growth_interval, _stream()), "zsv_scaler_update")

    def state(self) -> dict:
        """Host copy of the device state (synchronises): scale, growth_tracker, found_inf, steps_done."""
        host = self._state.cpu()
        return {"scale": float(host.view(torch.float32)[0]), "growth_tracker": int(host[1]), "found_inf": int(host[2]),
                "steps_done": int(host[3])}

    def get_scale(self) -> float:
        return self.state()["scale"]

    def state_dict(self) -> dict:
        """``GradScaler.state_dict()`` keys (``scale``, ``growth_factor``, ``backoff_factor``, ``growth_interval``,
        ``_growth_tracker``) plus ``steps_done`` -- the count of optimizer steps actually taken, which the Adam bias
        correction reads on the device.  One host sync."""
        st = self.state()
        return {"scale": st["scale"], "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": st["growth_tracker"], "steps_done": st["steps_done"]}

    def load_state_dict(self, state: dict) -> None:
        """Resume: uploads the 4-word device state.  A ``torch.amp.GradScaler`` state dict is accepted too (no
        ``steps_done``: the count is then taken from the optimizer when it adopts this scaler)."""
        self.growth_factor = float(state.get("growth_factor", self.growth_factor))
        self.backoff_factor = float(state.get("backoff_factor", self.backoff_factor))
        self.growth_interval = int(state.get("growth_interval", self.growth_interval))
        host = torch.zeros(4, dtype=torch.int32)
        host.view(torch.float32)[0] = float(state["scale"])
        host[1] = int(state.get("_growth_tracker", 0))
        host[3] = int(state.get("steps_done", 0))
        self._state.copy_(host)

    def _seed_steps_done(self, steps: int) -> None:
        host = self._state.cpu()
        host[3] = int(steps)
        self._state.copy_(host)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_buckets=None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._ring = [[None, None] for _ in range(4)]     # (pinned staging buffer, copy-done event)
        self._next = 0
        self.grad_buckets = grad_buckets                  # ddp.GradientSync (or None)
        self._static = None                               # (layout_version, table, count, chunks, [(p, grad ptr)], flats)
        self._host_steps = 0                              # steps taken without a scaler
        self._resumed = False                             # state came from load_state_dict (a scaler may then join late)
        self._scaler: Optional[LossScaler] = None

    # -- descriptor tables ------------------------------------------------------------------------
    def _ensure_state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @staticmethod
    def _check_param(p):
        if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
            raise RuntimeError("FusedAdam needs contiguous fp32 parameters on a HIP device (no CPU fallback)")

    def _upload(self, raw: bytearray, dev) -> torch.Tensor:
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
        return table

    def _dynamic_table(self, group):
        entries, first, keep = [], 0, []
        for p in group["params"]:
            if p.grad is None:
                continue
            self._check_param(p)
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            st = self._ensure_state(p)
            if self._scaler is None and int(st["step"].item()) != self._host_steps:
                raise RuntimeError("FusedAdam: every parameter that receives gradients must do so from the first step "
                                   "(one bias correction per launch)")
            n = p.numel()
            entries.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), n, first))
            keep.append(g)
            first += (n + _CHUNK - 1) // _CHUNK
        if not entries:
            return None
        raw = bytearray(b"".join(struct.pack("<QQQQqq", *e) for e in entries))
        table = self._upload(raw, keep[0].device)
        return table, len(entries), first, keep

    def _static_table(self, group):
        """Table over the flat gradient buckets, built once per bucket layout; None when it does not apply this
        step (buckets not built yet, several parameter groups, or a ``.grad`` that is not its bucket view)."""
        sync = self.grad_buckets
        if sync is None or not sync.ready or len(self.param_groups) != 1:
            return None
        if self._static is None or self._static[0] != sync.layout_version:
            mine = {id(p) for p in group["params"]}
            entries, first, expect, flats = [], 0, [], []
            for flat, rows in sync.bucket_layout():
                if not flat.is_cuda:
                    return None
                m_flat, v_flat = torch.zeros_like(flat), torch.zeros_like(flat)
                flats.append((m_flat, v_flat))
                for p, off in rows:
                    if id(p) not in mine:
                        raise RuntimeError("FusedAdam: a bucketed parameter is not in this optimizer")
                    self._check_param(p)
                    n = p.numel()
                    m, v = m_flat[off:off + n].view_as(p), v_flat[off:off + n].view_as(p)
                    st = self.state[p]
                    if st:                                   # state from the discovery step(s): move it into the flat buffers
                        m.copy_(st["exp_avg"])
                        v.copy_(st["exp_avg_sq"])
                    else:
                        st["step"] = torch.tensor(0.0)
                    st["exp_avg"], st["exp_avg_sq"] = m, v
                    gptr = flat.data_ptr() + 4 * off
                    entries.append((p.data_ptr(), gptr, m.data_ptr(), v.data_ptr(), n, first))
                    expect.append((p, gptr))
                    first += (n + _CHUNK - 1) // _CHUNK
            raw = bytearray(b"".join(struct.pack("<QQQQqq", *e) for e in entries))
            dev = expect[0][0].device
            table = torch.frombuffer(raw, dtype=torch.uint8).to(dev)          # once per layout: a blocking copy is fine
            self._static = (sync.layout_version, table, len(entries), first, expect, flats)
        _, table, count, chunks, expect, _ = self._static
        bucketed = {id(p) for p, _ in expect}
        for p, gptr in expect:
            if p.grad is None or p.grad.data_ptr() != gptr:
                return None
        for p in group["params"]:
            if p.grad is not None and id(p) not in bucketed:
                return None
        return table, count, chunks, []

    # -- step -------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None, scaler: Optional[LossScaler] = None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if scaler is not None and self._scaler is None:
            if self._host_steps:
                if not self._resumed:
                    raise RuntimeError("FusedAdam: a LossScaler must drive the optimizer from its first step "
                                       "(the count of steps taken lives in the scaler's device state)")
                # resumed from a checkpoint (load_state_dict): the scaler takes over the loaded step count unless its own
                # loaded state already carries one
                if scaler.state()["steps_done"] == 0:
                    scaler._seed_steps_done(self._host_steps)
                elif scaler.state()["steps_done"] != self._host_steps:
                    raise RuntimeError("FusedAdam: the loaded optimizer and scaler states disagree on the steps taken")
            self._scaler = scaler
        if self._scaler is not None and scaler is not self._scaler:
            raise RuntimeError("FusedAdam: this optimizer is driven by a LossScaler; step through scaler.step(optimizer)")
        lib = _lib.load()
        launched = False
        work = []
        for group in self.param_groups:
            built = self._static_table(group) or self._dynamic_table(group)
            if built is not None:
                work.append((group, built))
        if scaler is not None:
            # GradScaler.step checks EVERY gradient before the optimizer touches anything: an inf in the last group must
            # skip the first group's update too
            for _, (table, count, chunks, _keep) in work:
                with torch.cuda.device(table.device):
                    _lib.check(lib.zsv_grad_check_multi(table.data_ptr(), count, chunks, scaler.state_ptr, _stream()),
                               "zsv_grad_check_multi")
        for group, (table, count, chunks, keep) in work:
            lr, (b1, b2), eps = float(group["lr"]), group["betas"], float(group["eps"])
            with torch.cuda.device(table.device):
                if scaler is not None:
                    _lib.check(lib.zsv_adam_multi_scaled(table.data_ptr(), count, chunks, lr, float(b1), float(b2), eps,
                                                         scaler.state_ptr, _stream()), "zsv_adam_multi_scaled")
                else:
                    _lib.check(lib.zsv_adam_multi(table.data_ptr(), count, chunks, lr, float(b1), float(b2), eps,
                                                  self._host_steps + 1, _stream()), "zsv_adam_multi")
            # keep the uploaded table and any contiguous gradient copies alive until the stream is past the launch
            table.record_stream(torch.cuda.current_stream())
            del keep
            launched = True
        del work
        if launched:
            if scaler is None:
                self._host_steps += 1
                for group in self.param_groups:
                    for p in group["params"]:
                        if p.grad is not None:
                            self.state[p]["step"] += 1
            _lib.note_raw_write()                  # parameters updated through raw pointers
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = [int(st["step"].item()) for st in self.state.values() if "step" in st]
        if steps and min(steps) != max(steps):
            raise RuntimeError("FusedAdam: the loaded per-parameter step counts differ")
        self._host_steps = steps[0] if steps else 0
        self._resumed = True
        self._scaler = None
        for st in self.state.values():
            if "step" in st:
                st["step"] = st["step"].detach().to("cpu", torch.float32)
        self._static = None                        # the moments are re-homed into the flat buffers on the next step

    def state_dict(self):
        """With a ``LossScaler`` the number of steps actually taken is device state: fetched here (one sync)."""
        if self._scaler is not None:
            done = float(self._scaler.state()["steps_done"])
            for st in self.state.values():
                if "step" in st:
                    st["step"].fill_(done)
        return super().state_dict()
