"""Autograd bindings of the gfx950 kernels (``libzsv_hip.so``) -- the only compute path.

Every function here takes HIP-resident fp32 tensors, allocates outputs / workspaces with
torch (the C ABI never allocates) and launches on the current HIP stream.  Backward is a
``torch.autograd.Function`` per op calling the matching ``*_dgrad / *_wgrad / *_bwd`` entry
point.  A CPU tensor, a non-fp32 tensor or a missing library raises ``RuntimeError``.

What each op replaces in the reference is cited next to it.
"""
from __future__ import annotations

import os
import threading

from ctypes import byref, c_size_t, c_void_p
from typing import Optional, Sequence, Tuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import ConvDesc


def _triple(v) -> Tuple[int, int, int]:
    if isinstance(v, int):
        return (v, v, v)
    v = tuple(int(a) for a in v)
    if len(v) == 1:
        return (v[0],) * 3
    if len(v) != 3:
        raise ValueError(f"expected an int or 3 ints, got {v}")
    return v


def _require(*tensors: Optional[torch.Tensor]) -> None:
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "zeroshotvideoclassification_amd ops run only on an MI355X HIP device; got a "
                f"{t.device} tensor (there is no CPU fallback -- the CPU oracle lives in oracle/)")
        if t.dtype != torch.float32:
            raise RuntimeError(f"fp32 tensors expected, got {t.dtype}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _workspace(nbytes: int, device) -> Optional[torch.Tensor]:
    if nbytes <= 0:
        return None
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


# ---- weight gradients on a side stream --------------------------------------------------------------
# The weight gradient of a convolution is needed only by the optimizer step (or the data-parallel bucket it belongs to),
# while the input gradient is on the critical path of backward.  Every wgrad launch therefore goes to a side HIP stream
# (ZSV_WGRAD_STREAM=0 keeps it on the backward stream): the matrix-bound wgrad kernels then share the chip with the HBM-bound
# BatchNorm / pooling passes of the layers below instead of queueing in front of them (step 50.1 -> 47.7 ms on one device,
# profiles/r02_wgrad_side_stream_ab.txt; gradients are bit-identical, the kernels are the same).  The autograd engine runs a callback at the end of the backward pass that makes
# the calling stream wait for the side stream, so `.grad` is safe to read right after `loss.backward()`.
_WGRAD_SIDE = {}
_WGRAD_JOIN_QUEUED = {}
_WGRAD_SEEN = {}          # device index -> (graph task id, ids of the parameters that already produced a dw in that pass)
# Two private autograd entry points make the end-of-pass join possible; without them (another torch version) every wgrad
# launch is followed by an immediate wait on the calling stream -- correct, merely without the overlap.
_current_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)
_queue_engine_callback = getattr(getattr(torch.autograd.Variable, "_execution_engine", None), "queue_callback", None)


def wgrad_side_stream(device):
    """The side stream for weight gradients on ``device`` (None with ZSV_WGRAD_STREAM=0)."""
    if os.environ.get("ZSV_WGRAD_STREAM", "1") in ("", "0"):
        return None
    key = torch.device(device).index
    st = _WGRAD_SIDE.get(key)
    if st is None:
        st = _WGRAD_SIDE[key] = torch.cuda.Stream(device=device)
    return st


def join_wgrad_streams(stream=None):
    """Make ``stream`` (default: the current one) wait for every weight gradient launched so far."""
    for key, st in _WGRAD_SIDE.items():
        with torch.cuda.device(key):
            (stream or torch.cuda.current_stream()).wait_stream(st)


def _dw_read_early(param, device) -> bool:
    """Will something read (or add to) this weight gradient on the BACKWARD stream before the pass ends?
    * the parameter already holds a ``.grad`` (``zero_grad(set_to_none=False)``, gradient accumulation): autograd adds dw to
      it in place as soon as the backward function returns;
    * the weight is not a leaf: its gradient keeps flowing through the graph;
    * tensor hooks (``weight.register_hook``) run on dw right away;
    * post-accumulate hooks (``weight.register_post_accumulate_grad_hook``: optimizer-in-backward, clipping, logging) read
      ``weight.grad`` -- which IS dw -- on the backward stream as soon as it is set; a hook that joins the side stream itself
      carries ``_zsv_joins_wgrad = True`` (``ddp.GradientSync``'s does) and keeps the overlap;
    * the SAME weight produced a dw earlier in this pass (a module applied twice before backward, siamese / multi-clip
      forwards, tied weights): autograd's input buffer sums the two on the backward stream when the second one arrives --
      the first may still be in flight on the side stream."""
    if param is None:
        return False
    if not param.is_leaf or getattr(param, "_backward_hooks", None):
        return True
    post = getattr(param, "_post_accumulate_grad_hooks", None)
    if post and any(not getattr(h, "_zsv_joins_wgrad", False) for h in post.values()):
        return True
    if param.grad is not None:
        return True
    if _current_graph_task_id is None:
        return True                                    # cannot tell passes apart: be safe
    key = torch.device(device).index
    task = _current_graph_task_id()
    seen = _WGRAD_SEEN.get(key)
    if seen is None or seen[0] != task or task == -1:
        seen = _WGRAD_SEEN[key] = (task, set())
    pid = id(param)
    if pid in seen[1]:
        return True
    seen[1].add(pid)
    return False


def _on_wgrad_stream(launch, inputs, param=None):
    """Run ``launch(stream_handle) -> dw`` on the weight-gradient side stream (or the current stream when it is disabled).
    ``param``: the parameter the gradient belongs to; when its dw is consumed on the backward stream before the pass ends
    (``_dw_read_early``) that stream waits for the side stream right here."""
    dev = inputs[-1].device
    side = wgrad_side_stream(dev)
    if side is None:
        return launch(_stream())
    main = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(main)                                 # the operands are complete on the backward stream here
    side.wait_event(ready)
    with torch.cuda.stream(side):
        dw = launch(c_void_p(side.cuda_stream))        # (dw and the workspace come from the side stream's pool)
    for t in inputs:
        t.record_stream(side)                          # the side stream reads memory the backward stream owns
    dw.record_stream(main)                             # ... and the optimizer reads dw on the backward stream
    if getattr(_call_state, "defer_wgrad_join", False):
        return dw                                      # (the caller joins the side stream itself: hipGraph capture of a whole backward, amp.py)
    if _dw_read_early(param, dev) or not _queue_wgrad_join(dev):
        main.wait_stream(side)
    return dw


class deferred_wgrad_join:
    """Inside this context weight-gradient launches fork to the side stream and nobody waits for them: the caller joins once
    at the end (``join_wgrad_streams()``).  Used while a whole backward is captured into a hipGraph, where the per-pass
    autograd callback does not exist."""

    def __enter__(self):
        self.prev = getattr(_call_state, "defer_wgrad_join", False)
        _call_state.defer_wgrad_join = True
        return self

    def __exit__(self, *exc):
        _call_state.defer_wgrad_join = self.prev
        return False


def _queue_wgrad_join(device) -> bool:
    """Once per backward pass (autograd graph task) and device: a callback that runs when the pass ends and makes the
    stream the pass was launched on wait for the weight-gradient stream.  Keyed by the graph task id, so a pass that died
    with an exception cannot leave a stale "already queued" mark behind.  False: this torch has no such hook (the caller
    joins at once instead)."""
    if _current_graph_task_id is None or _queue_engine_callback is None:
        return False
    key = torch.device(device).index
    task = _current_graph_task_id()
    if task == -1:                                     # not inside an engine-driven pass (a Function.backward called by hand)
        return False
    if _WGRAD_JOIN_QUEUED.get(key) == task:
        return True
    _WGRAD_JOIN_QUEUED[key] = task
    main = torch.cuda.current_stream(device)
    side = _WGRAD_SIDE[key]
    _queue_engine_callback(lambda: main.wait_stream(side))
    return True


# ---- tap-validity tables of the weight-gradient kernels ---------------------------------------------------------------------
# zsv_conv3d_wgrad used to rebuild its per-voxel tap-validity table on every call (13 launches per R(2+1)D-18 step at the head
# of the weight-gradient queue's kernels).  The table depends on (input extents, kernel, padding) only: one
# per geometry and stream is kept here (a few hundred KB each) and handed to zsv_conv3d_wgrad_masked.
_WGRAD_MASKS = {}
_WGRAD_MASKS_GEN = [None]


def _wgrad_mask(d: ConvDesc, device, stream: c_void_p):
    if os.environ.get("ZSV_NO_WGRAD_MASK_CACHE"):
        return None
    if _WGRAD_MASKS_GEN[0] != _lib.knob_generation():        # the switches changed: which kernel serves a geometry may have too
        _WGRAD_MASKS.clear()
        _WGRAD_MASKS_GEN[0] = _lib.knob_generation()
    # (the table's CONTENT depends on extents / kernel / padding only; whether a geometry's kernel reads one depends on all of it)
    key = (torch.device(device).index, stream.value, _lib.knob_generation()) + tuple(getattr(d, f) for f, _ in ConvDesc._fields_)
    hit = _WGRAD_MASKS.get(key)
    if hit is not None:
        return hit if hit is not False else None
    lib = _lib.load()
    nbytes = int(lib.zsv_conv3d_wgrad_mask_bytes(byref(d)))
    if nbytes == 0:
        _WGRAD_MASKS[key] = False                       # this geometry's kernel reads no table
        return None
    mask = torch.empty(nbytes, dtype=torch.uint8, device=device)      # (allocated under the stream the launch runs on)
    _lib.check(lib.zsv_conv3d_wgrad_mask(byref(d), mask.data_ptr(), stream), "zsv_conv3d_wgrad_mask")
    _WGRAD_MASKS[key] = mask
    return mask


class KernelTimer:
    """Optional HIP-event timing of one kernel family on one geometry (used by bench.py for the
    live roofline figure): events are recorded on the launch stream around matching launches;
    nothing synchronises until ``durations_ms()`` is read."""

    def __init__(self, kind: str, geometry: dict):
        self.kind = kind
        self.geometry = dict(geometry)
        self.pairs = []

    def wants(self, kind: str, d: ConvDesc) -> bool:
        return kind == self.kind and all(getattr(d, k) == v for k, v in self.geometry.items())

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def stop(self, e0):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(torch.cuda.current_stream())
        self.pairs.append((e0, e1))

    def durations_ms(self):
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in self.pairs]


KERNEL_TIMER: Optional[KernelTimer] = None


# ---- weight panels kept across calls ------------------------------------------------------------------------------------
# Every convolution entry point re-lays its weights out ("packs a panel") in a launch of its own: 76 launches, 1.1-1.3 ms per
# R(2+1)D-18 training step (profiles/r03_pack_launches_cost.txt) for weights that change once per step.  Here every (weight,
# geometry, direction) keeps its panel in a buffer of its own; when a call finds its weight changed -- the first convolution of a
# step after the optimizer -- ALL stale panels (forward and input-gradient forms of every layer seen so far) are re-packed by ONE
# zsv_pack_multi launch on the calling stream, and the calls run the *_panel entry points, which skip their pack launch.
# "Changed" = the parameter's autograd version counter, its storage address, or `_lib.note_raw_write()` (kernels and `.data`
# writes that bypass the counter: FusedAdam, load_weights, broadcast_state -- anything else that writes `weight.data` directly must
# call `_lib.note_raw_write()` or `ops.invalidate_panels()`).  ZSV_NO_PANEL_CACHE=1 turns the cache off (every call packs, as before).
class _Panel:
    __slots__ = ("weight", "desc", "direction", "extras", "panel", "nbytes", "job", "version", "blocks")


class _PanelCache:
    def __init__(self, device):
        import threading
        self.device = device
        self.entries = {}            # key -> _Panel (None: this geometry has no single panel)
        self.knob_gen = _lib.knob_generation()
        self.table = None            # (tuple of entry keys, device table, total blocks, keep-alive host bytes)
        self.event = None            # recorded after the last multi-pack: consumers on another stream wait for it
        self.stream = None
        self.lock = threading.RLock()
        self.foreign = []            # streams other than the packing one that have read panels since the last pack
        self.repacks = 0             # multi-pack launches so far (tests / diagnostics)
        self.nbytes = 0              # bytes of all panels held
        # a panel exists per (weight, geometry): a caller that keeps changing clip shapes would grow the cache without bound, so it
        # is emptied (and refills with the shapes in use) once it holds more than this
        self.limit = int(os.environ.get("ZSV_PANEL_CACHE_MB", "8192")) << 20

    @staticmethod
    def _version(w):
        return (w._version, w.data_ptr(), _lib.raw_param_generation())

    def get(self, weight, d, direction, extras):
        """The up-to-date panel tensor of this call, or None (not cacheable / cache disabled)."""
        import weakref
        with self.lock:
            if self.knob_gen != _lib.knob_generation():          # the switches changed: layouts may have too
                self.entries.clear()
                self.nbytes = 0
                self.table = None
                self.knob_gen = _lib.knob_generation()
            key = (id(weight), direction, extras) + d.key
            e = self.entries.get(key, False)
            if e is not False and e is not None and e.weight() is not weight:
                e = False                                        # (the id was recycled by another tensor)
            if e is False:
                if self.nbytes > self.limit:
                    self.entries.clear()
                    self.nbytes = 0
                e = self._create(weight, d, direction, extras, weakref)
                self.entries[key] = e
                self.table = None
                if e is not None:
                    self.nbytes += e.nbytes
            if e is None:
                return None
            if e.version != self._version(weight):
                self._refresh()
            if os.environ.get("ZSV_PANEL_VERIFY"):
                self._verify(e, weight)
            cur = torch.cuda.current_stream(self.device)
            if self.stream is not None and cur != self.stream and self.event is not None:
                cur.wait_event(self.event)                       # packed on another stream
                e.panel.record_stream(cur)                       # ... which must also outlive this reader if the cache drops it
                if cur not in self.foreign:
                    self.foreign.append(cur)                     # (the next re-pack waits for what this stream has queued: ADVICE r3)
            return e.panel

    def _create(self, weight, d, direction, extras, weakref):
        lib = _lib.load()
        nb = c_size_t(0)
        _lib.check(lib.zsv_conv3d_panel_query(byref(d), direction, extras, byref(nb)), "zsv_conv3d_panel_query")
        if nb.value == 0:
            return None
        e = _Panel()
        e.weight = weakref.ref(weight)
        e.desc = ConvDesc(*d.key)
        e.direction, e.extras = direction, extras
        e.nbytes = int(nb.value)
        e.panel = torch.empty(e.nbytes, dtype=torch.uint8, device=weight.device)
        e.job = None
        e.version = None
        return e

    def _record(self, e, w):
        job = _lib.PackJob()
        _lib.check(_lib.load().zsv_conv3d_panel_job(byref(e.desc), e.direction, e.extras, w.data_ptr(), e.panel.data_ptr(), e.nbytes,
                                                   byref(job)), "zsv_conv3d_panel_job")
        e.job = job
        e.blocks = (int(job.total) + 1023) // 1024

    def _verify(self, e, weight):
        """ZSV_PANEL_VERIFY=1 (debug; one extra pack launch and a host sync per convolution call): pack this weight afresh into a
        scratch panel and compare it with the cached one.  A mismatch means the weight was written behind autograd's back
        (`.data` / `.detach()` alias, a raw kernel) without `_lib.note_raw_write()` / `ops.invalidate_panels()` -- forward and
        input gradient would silently have used the stale values."""
        scratch = torch.empty_like(e.panel)
        job = _lib.PackJob()
        lib = _lib.load()
        _lib.check(lib.zsv_conv3d_panel_job(byref(e.desc), e.direction, e.extras, weight.data_ptr(), scratch.data_ptr(), e.nbytes,
                                            byref(job)), "zsv_conv3d_panel_job")
        job.first_block = 0
        blocks = (int(job.total) + 1023) // 1024
        table = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(self.device)
        cur = torch.cuda.current_stream(self.device)
        if self.stream is not None and cur != self.stream and self.event is not None:
            cur.wait_event(self.event)
        with torch.cuda.device(self.device):
            _lib.check(lib.zsv_pack_multi(table.data_ptr(), 1, blocks, _stream()), "zsv_pack_multi")
        if not torch.equal(scratch, e.panel):
            raise RuntimeError(
                f"ZSV_PANEL_VERIFY: the cached weight panel of a {tuple(weight.shape)} convolution weight (direction {e.direction}) no "
                "longer matches the weight: it was modified without a version bump (a `.data` / `.detach()` alias, a raw kernel). "
                "Call zeroshotvideoclassification_amd.ops.invalidate_panels() or _lib.note_raw_write() after such writes.")

    def _refresh(self):
        """Re-pack every panel whose weight changed, in one launch on the current stream."""
        import ctypes
        stale, dead = [], []
        for key, e in self.entries.items():
            if e is None:
                continue
            w = e.weight()
            if w is None:
                dead.append(key)
                continue
            v = self._version(w)
            if e.version != v:
                if e.job is None or e.job.w != w.data_ptr():
                    self._record(e, w)                           # (first use, or the storage moved)
                    self.table = None
                stale.append((key, e, v))
        for key in dead:
            self.nbytes -= self.entries[key].nbytes
            del self.entries[key]
            self.table = None
        if not stale:
            return
        keys = tuple(k for k, _, _ in stale)
        if self.table is None or self.table[0] != keys:
            first, raw = 0, bytearray()
            for _, e, _ in stale:
                e.job.first_block = first
                first += e.blocks
                raw += bytes(e.job)
            host = torch.frombuffer(raw, dtype=torch.uint8)
            self.table = (keys, host.to(self.device), first)      # (a blocking copy: only when the set of stale panels changes)
        _, table, blocks = self.table
        packing = torch.cuda.current_stream(self.device)
        for st in self.foreign:                                  # readers of the old panels on other streams finish first
            if st != packing:
                packing.wait_stream(st)
        if self.stream is not None and self.stream != packing:
            packing.wait_stream(self.stream)                     # ... and so do the previous packing stream's own readers
        self.foreign = []
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().zsv_pack_multi(table.data_ptr(), len(stale), blocks, _stream()), "zsv_pack_multi")
        self.stream = torch.cuda.current_stream(self.device)
        self.event = torch.cuda.Event()
        self.event.record(self.stream)
        self.repacks += 1
        for _, e, v in stale:
            e.version = v


_PANEL_CACHES = {}


_call_state = threading.local()       # .recording: autograd's grad mode at the public entry point (a Function's forward always runs without it)


def _recording() -> bool:
    return getattr(_call_state, "recording", True)


def _panel_for(weight, d, direction, extras, recording=True):
    """``recording``: the call is part of an autograd graph (a forward whose inputs need gradients, or any backward)."""
    if os.environ.get("ZSV_NO_PANEL_CACHE"):
        return None
    # Without autograd (inference, evaluation) every call packs its own panel, as before: a change of `weight` is seen through
    # its version counter, and writes through a `.data` alias (EMA copies, pruning, hand-made weight edits -- the things people do
    # to an eval model) bump none.  In a training loop the next forward after `optimizer.step()` re-packs from whatever the weights
    # hold by then, so `.data` edits between the step and that forward are picked up.  ZSV_PANEL_CACHE_EVAL=1 caches there too.
    if not recording and not os.environ.get("ZSV_PANEL_CACHE_EVAL"):
        return None
    key = weight.device.index
    cache = _PANEL_CACHES.get(key)
    if cache is None:
        cache = _PANEL_CACHES[key] = _PanelCache(weight.device)
    return cache.get(weight, d, direction, 1 if extras else 0)


def invalidate_panels():
    """Forget every cached weight panel (after writing weights behind autograd's back without ``_lib.note_raw_write()``)."""
    _PANEL_CACHES.clear()


def conv_desc(x_shape: Sequence[int], w_shape: Sequence[int], stride, padding) -> ConvDesc:
    n, cin, ti, hi, wi = (int(v) for v in x_shape)
    cout, cin_w, kt, kh, kw = (int(v) for v in w_shape)
    if cin != cin_w:
        raise RuntimeError(f"conv3d: input has {cin} channels, weight expects {cin_w}")
    st, sh, sw = _triple(stride)
    pt, ph, pw = _triple(padding)
    to = (ti + 2 * pt - kt) // st + 1
    ho = (hi + 2 * ph - kh) // sh + 1
    wo = (wi + 2 * pw - kw) // sw + 1
    if min(to, ho, wo) <= 0:
        raise RuntimeError(f"conv3d: kernel {(kt, kh, kw)} does not fit input {(ti, hi, wi)}")
    d = ConvDesc(n, cin, ti, hi, wi, cout, to, ho, wo, kt, kh, kw, st, sh, sw, pt, ph, pw)
    d.key = (n, cin, ti, hi, wi, cout, to, ho, wo, kt, kh, kw, st, sh, sw, pt, ph, pw)      # (hashable twin: panel cache)
    return d


class SkipLink:
    """Joins the two gradient paths of an identity shortcut (`out += residual` with residual = the
    block input, resnet.py:102-110).  The block's tail BatchNorm+add+ReLU backward runs first and parks
    the shortcut gradient here instead of returning it; the backward of the block's first convolution
    (whose input IS that residual) adds it in its dgrad epilogue -- one pass less over the block input."""
    __slots__ = ("dres", "armed")

    def __init__(self):
        self.dres = None
        self.armed = False        # set by the convolution's forward: its backward will collect `dres`


class DownLink:
    """Joins the two gradient paths of a block with a strided 1x1x1 shortcut convolution (`downsample`, resnet.py:240-246): the
    block input feeds the strided first convolution AND the shortcut convolution.  The shortcut's backward runs first (it was
    recorded later) and parks its input gradient here in COMPACT form -- only the voxels it reads -- instead of returning a
    zero-filled full-size tensor; the backward of the block's first convolution adds it in its dgrad epilogue
    (zsv_conv3d_dgrad_add_strided).  If the order ever differs (`consumed` already set) the shortcut returns its gradient the
    ordinary way."""
    __slots__ = ("strides", "dsub", "armed", "consumed")

    def __init__(self, strides):
        self.strides = tuple(int(v) for v in strides)
        self.dsub = None
        self.armed = False        # set by the first convolution's forward: its backward will collect `dsub`
        self.consumed = False


# ------------------------------------------------------------------------------------------
class _Conv3d(Function):
    """aten::conv3d fwd / dgrad / wgrad (resnet.py:23-30,40-52,63-70,170,181,184,270;
    network.py:102-117), optional bias and fused ReLU (network.py:147-162)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, relu, want_stats, skip_link=None, down_link=None, down_src=None):
        _require(x, weight, bias)
        ctx.set_materialize_grads(False)      # no zero tensor for the (non-differentiable) statistics output
        if x.dim() != 5 or weight.dim() != 5:
            raise RuntimeError("conv3d expects (N,C,T,H,W) input and (O,I,kT,kH,kW) weight")
        x = x.contiguous()
        weight = weight.contiguous()
        d = conv_desc(x.shape, weight.shape, stride, padding)
        y = torch.empty((d.N, d.Cout, d.To, d.Ho, d.Wo), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        timer = KERNEL_TIMER
        with torch.cuda.device(x.device):
            nbytes = lib.zsv_conv3d_fwd_workspace_bytes(byref(d))
            ws = _workspace(nbytes, x.device)
            # BatchNorm partial statistics from the epilogue (consumed by the BatchNorm that follows)
            tiles = 0
            stats = None
            if want_stats and bias is None and not relu:
                tiles = lib.zsv_conv3d_fwd_stat_tiles(byref(d), y.data_ptr())
                if tiles > 0:
                    stats = torch.empty((2, d.Cout, tiles), dtype=torch.float32, device=x.device)
            ev = timer.start() if (timer is not None and timer.wants("conv_fwd", d)) else None
            panel = _panel_for(weight, d, 0, bias is not None or relu or stats is not None, _recording())
            if panel is None:
                _lib.check(lib.zsv_conv3d_fwd_stats(byref(d), x.data_ptr(), weight.data_ptr(), _ptr(bias), y.data_ptr(),
                                                    1 if relu else 0, _ptr(stats), tiles, _ptr(ws), nbytes, _stream()),
                           "zsv_conv3d_fwd")
            else:
                _lib.check(lib.zsv_conv3d_fwd_full_panel(byref(d), x.data_ptr(), weight.data_ptr(), _ptr(bias), None, y.data_ptr(),
                                                         1 if relu else 0, _ptr(stats), tiles, _ptr(ws), nbytes, _stream(),
                                                         panel.data_ptr(), panel.numel()), "zsv_conv3d_fwd (panel)")
            if ev is not None:
                timer.stop(ev)
        ctx.desc = d
        ctx.relu = bool(relu)
        ctx.has_bias = bias is not None
        ctx.skip_link = None
        if skip_link is not None and lib.zsv_conv3d_dgrad_add_supported(byref(d)):
            ctx.skip_link = skip_link
            skip_link.armed = True
        ctx.down_link = None          # this convolution collects the strided shortcut's gradient in its dgrad
        if down_link is not None and lib.zsv_conv3d_dgrad_add_strided_supported(byref(d), *down_link.strides):
            ctx.down_link = down_link
            down_link.armed = True
        ctx.down_src = None           # this convolution IS the strided 1x1x1 shortcut: park the compact gradient
        if down_src is not None and down_src.armed and (d.kT, d.kH, d.kW) == (1, 1, 1) and (d.pT, d.pH, d.pW) == (0, 0, 0) \
                and (d.sT, d.sH, d.sW) == down_src.strides:
            ctx.down_src = down_src
        ctx.save_for_backward(x, weight, y if relu else None)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _dstats=None):
        x, weight, y = ctx.saved_tensors
        d = ctx.desc
        lib = _lib.load()
        if dy is None:                          # output unused downstream
            return (None,) * 10
        dy = dy.contiguous()
        dx = dw = db = None
        with torch.cuda.device(dy.device):
            if ctx.relu:
                g = torch.empty_like(dy)
                if ctx.has_bias and ctx.needs_input_grad[2]:
                    # ReLU mask and bias gradient in one pass over dy (C3D: every convolution has both)
                    n_, c_ = int(dy.shape[0]), int(dy.shape[1])
                    s_ = dy.numel() // (n_ * c_)
                    db = torch.empty(c_, dtype=torch.float32, device=dy.device)
                    nb = lib.zsv_channel_sum_workspace_bytes(n_, c_, s_)
                    wsb = _workspace(nb, dy.device)
                    _lib.check(lib.zsv_relu_bwd_bias(dy.data_ptr(), y.data_ptr(), g.data_ptr(), n_, c_, s_, db.data_ptr(),
                                                     _ptr(wsb), nb, _stream()), "zsv_relu_bwd_bias")
                else:
                    _lib.check(lib.zsv_relu_bwd(dy.data_ptr(), y.data_ptr(), g.data_ptr(), g.numel(), _stream()), "zsv_relu_bwd")
                dy = g
            src = ctx.down_src
            if ctx.needs_input_grad[0] and src is not None and not src.consumed:
                # strided 1x1x1 shortcut: its input gradient over its own output voxels only (a stride-1 problem on the compact
                # grid), parked for the block's first convolution; this node contributes no full-size gradient
                d1 = conv_desc((d.N, d.Cin, d.To, d.Ho, d.Wo), weight.shape, (1, 1, 1), (0, 0, 0))
                dsub = torch.empty((d.N, d.Cin, d.To, d.Ho, d.Wo), dtype=torch.float32, device=dy.device)
                nbytes = lib.zsv_conv3d_dgrad_workspace_bytes(byref(d1))
                ws = _workspace(nbytes, dy.device)
                panel = _panel_for(weight, d1, 1, False)
                if panel is None:
                    _lib.check(lib.zsv_conv3d_dgrad(byref(d1), dy.data_ptr(), weight.data_ptr(), dsub.data_ptr(), _ptr(ws), nbytes,
                                                    _stream()), "zsv_conv3d_dgrad (compact shortcut gradient)")
                else:
                    _lib.check(lib.zsv_conv3d_dgrad_add_panel(byref(d1), dy.data_ptr(), weight.data_ptr(), None, dsub.data_ptr(), _ptr(ws),
                                                              nbytes, _stream(), panel.data_ptr(), panel.numel()),
                               "zsv_conv3d_dgrad (compact shortcut gradient, panel)")
                src.dsub = dsub
            elif ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                nbytes = lib.zsv_conv3d_dgrad_workspace_bytes(byref(d))
                ws = _workspace(nbytes, dy.device)
                add = None
                if ctx.skip_link is not None:            # the shortcut's gradient, parked by the block's tail
                    add, ctx.skip_link.dres = ctx.skip_link.dres, None
                link = ctx.down_link
                if link is not None:
                    link.consumed = True
                panel = _panel_for(weight, d, 1, False)
                pp, pn = (panel.data_ptr(), panel.numel()) if panel is not None else (None, 0)
                if link is not None and link.dsub is not None:      # the strided shortcut's compact gradient
                    sub, link.dsub = link.dsub, None
                    if panel is None:
                        _lib.check(lib.zsv_conv3d_dgrad_add_strided(byref(d), dy.data_ptr(), weight.data_ptr(), sub.data_ptr(),
                                                                    link.strides[0], link.strides[1], link.strides[2], dx.data_ptr(),
                                                                    _ptr(ws), nbytes, _stream()), "zsv_conv3d_dgrad_add_strided")
                    else:
                        _lib.check(lib.zsv_conv3d_dgrad_add_strided_panel(byref(d), dy.data_ptr(), weight.data_ptr(), sub.data_ptr(),
                                                                          link.strides[0], link.strides[1], link.strides[2], dx.data_ptr(),
                                                                          _ptr(ws), nbytes, _stream(), pp, pn),
                                   "zsv_conv3d_dgrad_add_strided (panel)")
                elif panel is None:
                    _lib.check(lib.zsv_conv3d_dgrad_add(byref(d), dy.data_ptr(), weight.data_ptr(), _ptr(add), dx.data_ptr(),
                                                        _ptr(ws), nbytes, _stream()), "zsv_conv3d_dgrad")
                else:
                    _lib.check(lib.zsv_conv3d_dgrad_add_panel(byref(d), dy.data_ptr(), weight.data_ptr(), _ptr(add), dx.data_ptr(),
                                                              _ptr(ws), nbytes, _stream(), pp, pn), "zsv_conv3d_dgrad (panel)")
            if ctx.needs_input_grad[1]:
                nbytes = lib.zsv_conv3d_wgrad_workspace_bytes(byref(d))

                def launch(stream):
                    out = torch.empty_like(weight)
                    ws = _workspace(nbytes, dy.device)
                    mask = _wgrad_mask(d, dy.device, stream)         # geometry-only table, built once per (geometry, stream)
                    _lib.check(lib.zsv_conv3d_wgrad_masked(byref(d), x.data_ptr(), dy.data_ptr(), out.data_ptr(), _ptr(ws), nbytes,
                                                           _ptr(mask), stream), "zsv_conv3d_wgrad")
                    return out

                dw = _on_wgrad_stream(launch, (x, dy), weight)
            if ctx.has_bias and ctx.needs_input_grad[2] and db is None:
                db = channel_sum(dy)
        return dx, dw, db, None, None, None, None, None, None, None


def conv3d(x, weight, bias=None, stride=1, padding=0, relu=False, want_stats=False):
    """``want_stats``: also return the epilogue's BatchNorm partial statistics (or None when this
    geometry does not produce them) as ``(y, stats)``.  An input tagged with a ``SkipLink`` (by
    ``BasicBlock.forward``: the tensor is also the block's identity shortcut) makes this convolution's
    backward add the shortcut gradient in its dgrad epilogue."""
    link = x.__dict__.pop("_zsv_skip_link", None) if hasattr(x, "__dict__") else None
    down = x.__dict__.pop("_zsv_down_link", None) if hasattr(x, "__dict__") else None        # (``DownLink``: consumer / producer tags)
    src = x.__dict__.pop("_zsv_down_src", None) if hasattr(x, "__dict__") else None
    if not (torch.is_grad_enabled() and x.requires_grad):
        link = down = src = None
    # "recording" = a graph is really being built through this call: a forward under grad mode whose operands need no
    # gradient (frozen `fixconvs` trunk, feature extraction without no_grad) is inference as far as the panel cache is concerned
    _call_state.recording = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad))
    y, stats = _Conv3d.apply(x, weight, bias, _triple(stride), _triple(padding), bool(relu), bool(want_stats), link, down, src)
    return (y, stats) if want_stats else y


def channel_sum(t: torch.Tensor) -> torch.Tensor:
    """sum over every axis but the channel axis 1 of an (N, C, ...) tensor (bias gradients)."""
    _require(t)
    t = t.contiguous()
    n, c = int(t.shape[0]), int(t.shape[1])
    s = t.numel() // (n * c)
    lib = _lib.load()
    out = torch.empty(c, dtype=torch.float32, device=t.device)
    nbytes = lib.zsv_channel_sum_workspace_bytes(n, c, s)
    ws = _workspace(nbytes, t.device)
    with torch.cuda.device(t.device):
        _lib.check(lib.zsv_channel_sum(t.data_ptr(), n, c, s, out.data_ptr(), _ptr(ws), nbytes, _stream()),
                   "zsv_channel_sum")
    return out


# ------------------------------------------------------------------------------------------
class _BatchNormAct(Function):
    """nn.BatchNorm3d (resnet.py:48,95,97,183,186,272) + optional `out += residual`
    (resnet.py:110) + optional ReLU (resnet.py:49,95,111) in one pass."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, training, momentum, eps, relu, stats,
                skip_link=None):
        _require(x, gamma, beta, running_mean, running_var, residual, stats)
        x = x.contiguous()
        n, c = int(x.shape[0]), int(x.shape[1])
        s = x.numel() // (n * c)
        if residual is not None:
            if residual.shape != x.shape:
                raise RuntimeError(f"residual shape {tuple(residual.shape)} != {tuple(x.shape)}")
            residual = residual.contiguous()
        lib = _lib.load()
        y = torch.empty_like(x)
        nbytes = lib.zsv_bn_workspace_bytes(n, c, s)
        ws = _workspace(nbytes, x.device)
        ctx.training = bool(training)
        with torch.cuda.device(x.device):
            if training:
                save_mean = torch.empty(c, dtype=torch.float32, device=x.device)
                save_invstd = torch.empty(c, dtype=torch.float32, device=x.device)
                tiles = 0
                if stats is not None:
                    if stats.dim() != 3 or stats.shape[0] != 2 or stats.shape[1] != c or not stats.is_contiguous():
                        raise RuntimeError("conv statistics do not match this BatchNorm")
                    tiles = int(stats.shape[2])
                _lib.check(lib.zsv_bn_fwd_train_stats(x.data_ptr(), n, c, s, _ptr(gamma), _ptr(beta), _ptr(residual),
                                                      1 if relu else 0, y.data_ptr(), save_mean.data_ptr(),
                                                      save_invstd.data_ptr(), _ptr(running_mean), _ptr(running_var),
                                                      float(momentum), float(eps), _ptr(stats), tiles, _ptr(ws), nbytes,
                                                      _stream()), "zsv_bn_fwd_train")
                if running_mean is not None:
                    _lib.note_raw_write(parameters=False)          # running statistics written behind the version counters
            else:
                if running_mean is None or running_var is None:
                    raise RuntimeError("eval-mode BatchNorm needs running statistics")
                save_mean = save_invstd = None
                _lib.check(lib.zsv_bn_fwd_eval(x.data_ptr(), n, c, s, _ptr(gamma), _ptr(beta), running_mean.data_ptr(),
                                               running_var.data_ptr(), _ptr(residual), 1 if relu else 0, float(eps),
                                               y.data_ptr(), _ptr(ws), nbytes, _stream()), "zsv_bn_fwd_eval")
        ctx.relu = bool(relu)
        ctx.has_res = residual is not None
        ctx.dims = (n, c, s)
        ctx.skip_link = skip_link if (skip_link is not None and skip_link.armed and residual is not None) else None
        # with a ReLU but no residual the backward recomputes the mask from x: y need not be kept
        ctx.save_for_backward(x, gamma, beta, save_mean, save_invstd, y if (relu and residual is not None) else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("backward through eval-mode BatchNorm is not part of the hot path "
                               "(the reference evaluates under torch.no_grad(), main.py:230)")
        x, gamma, beta, save_mean, save_invstd, y = ctx.saved_tensors
        n, c, s = ctx.dims
        lib = _lib.load()
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
        want_res = ctx.has_res and ctx.needs_input_grad[5]
        # without a fused ReLU the residual gradient is dy itself
        dres = torch.empty_like(x) if (want_res and ctx.relu) else None
        nbytes = lib.zsv_bn_workspace_bytes(n, c, s)
        ws = _workspace(nbytes, x.device)
        with torch.cuda.device(x.device):
            mode = 0 if not ctx.relu else (1 if ctx.has_res else 2)
            _lib.check(lib.zsv_bn_bwd(dy.data_ptr(), x.data_ptr(), _ptr(y), n, c, s, _ptr(gamma), _ptr(beta),
                                      save_mean.data_ptr(), save_invstd.data_ptr(), mode, dx.data_ptr(), _ptr(dres),
                                      dgamma.data_ptr(), dbeta.data_ptr(), _ptr(ws), nbytes, _stream()), "zsv_bn_bwd")
        if want_res and not ctx.relu:
            dres = dy
        if want_res and ctx.skip_link is not None:
            # identity shortcut: the first convolution of the block adds this in its dgrad epilogue
            ctx.skip_link.dres = dres
            dres = None
        return (dx if ctx.needs_input_grad[0] else None, dgamma if ctx.needs_input_grad[1] else None,
                dbeta if ctx.needs_input_grad[2] else None, None, None, dres if want_res else None,
                None, None, None, None, None, None)


def batch_norm_act(x, gamma, beta, running_mean, running_var, residual=None, training=True, momentum=0.1,
                   eps=1e-5, relu=False, stats=None, skip_link=None):
    """``stats``: BatchNorm partial statistics of ``x`` from the producing convolution's epilogue
    (``conv3d(..., want_stats=True)``); the kernel then skips its own pass over ``x``."""
    return _BatchNormAct.apply(x, gamma, beta, running_mean, running_var, residual, bool(training), float(momentum),
                               float(eps), bool(relu), stats if training else None, skip_link)


# ------------------------------------------------------------------------------------------
# BatchNorm + ReLU folded into the convolution that consumes it (Conv2Plus1D's mid tensor, resnet.py:46-52)
class _BatchNormDeferred(Function):
    """Training-mode ``BatchNorm3d`` up to, but without, the normalise pass: returns its input (as the handle the
    consuming convolution differentiates against) and ``coef`` = per-channel (scale, shift).  The consumer
    (``conv3d_pre``) reads ``relu(x * scale + shift)`` on the fly; its input gradient is the gradient w.r.t. that virtual
    activation, which is exactly what this backward expects (``zsv_bn_bwd`` with the ReLU mask recomputed from x)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, stats):
        ctx.set_materialize_grads(False)
        _require(x, gamma, beta, running_mean, running_var, stats)
        if not x.is_contiguous():
            raise RuntimeError("deferred BatchNorm needs a contiguous input")
        n, c = int(x.shape[0]), int(x.shape[1])
        s = x.numel() // (n * c)
        lib = _lib.load()
        pitch = (c + 15) // 16 * 16
        coef = torch.empty((2, pitch), dtype=torch.float32, device=x.device)
        save_mean = torch.empty(c, dtype=torch.float32, device=x.device)
        save_invstd = torch.empty(c, dtype=torch.float32, device=x.device)
        tiles = 0
        if stats is not None:
            if stats.dim() != 3 or stats.shape[0] != 2 or stats.shape[1] != c or not stats.is_contiguous():
                raise RuntimeError("conv statistics do not match this BatchNorm")
            tiles = int(stats.shape[2])
        nbytes = lib.zsv_bn_workspace_bytes(n, c, s)
        ws = _workspace(nbytes, x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.zsv_bn_fwd_train_coeffs(x.data_ptr(), n, c, s, _ptr(gamma), _ptr(beta), save_mean.data_ptr(),
                                                   save_invstd.data_ptr(), _ptr(running_mean), _ptr(running_var),
                                                   float(momentum), float(eps), _ptr(stats), tiles, coef.data_ptr(), pitch,
                                                   _ptr(ws), nbytes, _stream()), "zsv_bn_fwd_train_coeffs")
        if running_mean is not None:
            _lib.note_raw_write(parameters=False)
        ctx.dims = (n, c, s)
        ctx.save_for_backward(x, gamma, beta, save_mean, save_invstd)
        ctx.mark_non_differentiable(coef)
        return x.view_as(x), coef

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _dcoef=None):
        x, gamma, beta, save_mean, save_invstd = ctx.saved_tensors
        n, c, s = ctx.dims
        lib = _lib.load()
        if g is None:                           # output unused downstream
            return (None,) * 8
        g = g.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
        nbytes = lib.zsv_bn_workspace_bytes(n, c, s)
        ws = _workspace(nbytes, x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.zsv_bn_bwd(g.data_ptr(), x.data_ptr(), None, n, c, s, _ptr(gamma), _ptr(beta), save_mean.data_ptr(),
                                      save_invstd.data_ptr(), 2, dx.data_ptr(), None, dgamma.data_ptr(), dbeta.data_ptr(),
                                      _ptr(ws), nbytes, _stream()), "zsv_bn_bwd")
        return (dx if ctx.needs_input_grad[0] else None, dgamma if ctx.needs_input_grad[1] else None,
                dbeta if ctx.needs_input_grad[2] else None, None, None, None, None, None)


class _Conv3dPre(Function):
    """``conv3d(relu(x * scale + shift), w)`` with the affine + ReLU applied inside the kernels (forward: the direct kernel's
    PRE form; weight gradient: the frame-ring kernel's PRE form)."""

    @staticmethod
    def forward(ctx, x, coef, weight, stride, padding, want_stats):
        ctx.set_materialize_grads(False)
        _require(x, coef, weight)
        weight = weight.contiguous()
        d = conv_desc(x.shape, weight.shape, stride, padding)
        lib = _lib.load()
        if not lib.zsv_conv3d_pre_supported(byref(d)):
            raise RuntimeError("conv3d_pre: this geometry has no fused BatchNorm path (check conv_pre_supported first)")
        y = torch.empty((d.N, d.Cout, d.To, d.Ho, d.Wo), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            nbytes = lib.zsv_conv3d_fwd_workspace_bytes(byref(d))
            ws = _workspace(nbytes, x.device)
            tiles, stats = 0, None
            if want_stats:
                tiles = lib.zsv_conv3d_fwd_stat_tiles(byref(d), y.data_ptr())
                if tiles > 0:
                    stats = torch.empty((2, d.Cout, tiles), dtype=torch.float32, device=x.device)
            panel = _panel_for(weight, d, 0, False, _recording())
            if panel is None:
                _lib.check(lib.zsv_conv3d_fwd_pre(byref(d), x.data_ptr(), coef.data_ptr(), int(coef.shape[1]), weight.data_ptr(),
                                                  y.data_ptr(), _ptr(stats), tiles, _ptr(ws), nbytes, _stream()), "zsv_conv3d_fwd_pre")
            else:
                _lib.check(lib.zsv_conv3d_fwd_pre_panel(byref(d), x.data_ptr(), coef.data_ptr(), int(coef.shape[1]), weight.data_ptr(),
                                                        y.data_ptr(), _ptr(stats), tiles, _ptr(ws), nbytes, _stream(), panel.data_ptr(),
                                                        panel.numel()), "zsv_conv3d_fwd_pre (panel)")
        ctx.desc = d
        ctx.save_for_backward(x, coef, weight)
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        return y, stats

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _dstats=None):
        x, coef, weight = ctx.saved_tensors
        d = ctx.desc
        lib = _lib.load()
        if dy is None:
            return None, None, None, None, None, None
        dy = dy.contiguous()
        dx = dw = None
        with torch.cuda.device(dy.device):
            if ctx.needs_input_grad[0]:                     # gradient w.r.t. the virtual activation relu(bn(x))
                dx = torch.empty_like(x)
                nbytes = lib.zsv_conv3d_dgrad_workspace_bytes(byref(d))
                ws = _workspace(nbytes, dy.device)
                panel = _panel_for(weight, d, 1, False)
                if panel is None:
                    _lib.check(lib.zsv_conv3d_dgrad(byref(d), dy.data_ptr(), weight.data_ptr(), dx.data_ptr(), _ptr(ws), nbytes,
                                                    _stream()), "zsv_conv3d_dgrad")
                else:
                    _lib.check(lib.zsv_conv3d_dgrad_add_panel(byref(d), dy.data_ptr(), weight.data_ptr(), None, dx.data_ptr(), _ptr(ws),
                                                              nbytes, _stream(), panel.data_ptr(), panel.numel()), "zsv_conv3d_dgrad (panel)")
            if ctx.needs_input_grad[2]:
                nbytes = lib.zsv_conv3d_wgrad_workspace_bytes(byref(d))
                pitch = int(coef.shape[1])

                def launch(stream):
                    out = torch.empty_like(weight)
                    ws = _workspace(nbytes, dy.device)
                    _lib.check(lib.zsv_conv3d_wgrad_pre(byref(d), x.data_ptr(), coef.data_ptr(), pitch, dy.data_ptr(),
                                                        out.data_ptr(), _ptr(ws), nbytes, stream), "zsv_conv3d_wgrad_pre")
                    return out

                dw = _on_wgrad_stream(launch, (x, coef, dy), weight)
        return dx, None, dw, None, None, None


def conv_pre_supported(x_shape, weight_shape, stride, padding) -> bool:
    d = conv_desc(x_shape, weight_shape, stride, padding)
    return bool(_lib.load().zsv_conv3d_pre_supported(byref(d)))


def bn_module_deferred(x, bn: torch.nn.Module, stats=None):
    """``bn`` in training mode without its normalise pass: ``(x_handle, coef)`` for ``conv3d_pre`` (running statistics
    and ``num_batches_tracked`` are updated exactly as ``bn_module_act`` does)."""
    if bn.momentum is None or not bn.training:
        raise RuntimeError("deferred BatchNorm: training mode with a momentum only")
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        if _NBT_PENDING is not None:
            _NBT_PENDING.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BatchNormDeferred.apply(x, bn.weight, bn.bias, rm, rv, float(bn.momentum), float(bn.eps), stats)


def conv3d_pre(x, coef, weight, stride=1, padding=0, want_stats=False):
    _call_state.recording = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or coef.requires_grad)
    y, stats = _Conv3dPre.apply(x, coef, weight, _triple(stride), _triple(padding), bool(want_stats))
    return (y, stats) if want_stats else y


_NBT_PENDING = None      # list of num_batches_tracked buffers to increment when the enclosing forward ends


class batched_bn_counters:
    """Within this context the ``num_batches_tracked += 1`` of every training-mode BatchNorm
    (37 one-element kernels in R(2+1)D-18) is deferred and applied as one ``_foreach_add_`` on exit."""

    def __enter__(self):
        global _NBT_PENDING
        self.outer = _NBT_PENDING
        if self.outer is None:
            _NBT_PENDING = []
        return self

    def __exit__(self, *exc):
        global _NBT_PENDING
        if self.outer is None:
            pending, _NBT_PENDING = _NBT_PENDING, None
            if pending:
                torch._foreach_add_(pending, 1)
        return False


def bn_module_act(x, bn: torch.nn.Module, residual=None, relu=False, stats=None, skip_link=None):
    """Apply an ``nn.BatchNorm3d``-like module's parameters through the fused kernel, with
    torch's train/eval and running-statistics semantics (momentum=None -> cumulative average
    is not used anywhere in the reference and is rejected)."""
    if bn.momentum is None:
        raise RuntimeError("cumulative-average BatchNorm (momentum=None) is not supported")
    use_batch_stats = bn.training or (bn.running_mean is None and bn.running_var is None)
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if _NBT_PENDING is not None:
            _NBT_PENDING.append(bn.num_batches_tracked)         # one fused increment per forward
        else:
            bn.num_batches_tracked.add_(1)
    rm = bn.running_mean if (bn.track_running_stats or not use_batch_stats) else None
    rv = bn.running_var if (bn.track_running_stats or not use_batch_stats) else None
    return batch_norm_act(x, bn.weight, bn.bias, rm, rv, residual, use_batch_stats, bn.momentum, bn.eps, relu, stats,
                          skip_link)


# ------------------------------------------------------------------------------------------
class _ReLU(Function):
    """nn.ReLU / F.relu (resnet.py:49,95,98)."""

    @staticmethod
    def forward(ctx, x):
        _require(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().zsv_relu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "zsv_relu_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        with torch.cuda.device(dy.device):
            _lib.check(_lib.load().zsv_relu_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), dy.numel(), _stream()),
                       "zsv_relu_bwd")
        return dx


def relu(x):
    return _ReLU.apply(x)


class _AddReLU(Function):
    """`out += residual; out = relu(out)` (resnet.py:110-111) when no BatchNorm precedes it."""

    @staticmethod
    def forward(ctx, a, b):
        _require(a, b)
        if a.shape != b.shape:
            raise RuntimeError("add_relu: shapes differ")
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        with torch.cuda.device(a.device):
            _lib.check(_lib.load().zsv_add_relu_fwd(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream()),
                       "zsv_add_relu_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        g = torch.empty_like(dy)
        with torch.cuda.device(dy.device):
            _lib.check(_lib.load().zsv_relu_bwd(dy.data_ptr(), y.data_ptr(), g.data_ptr(), dy.numel(), _stream()),
                       "zsv_relu_bwd")
        return g, g


def add_relu(a, b):
    return _AddReLU.apply(a, b)


# ------------------------------------------------------------------------------------------
class _MeanPool(Function):
    """torch.mean(f, dim=(2,3,4)) (network.py:595) == AdaptiveAvgPool3d(1).flatten(1) (resnet.py:251-253)."""

    @staticmethod
    def forward(ctx, x):
        _require(x)
        x = x.contiguous()
        n, c = int(x.shape[0]), int(x.shape[1])
        s = x.numel() // (n * c)
        y = torch.empty((n, c), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().zsv_meanpool_fwd(x.data_ptr(), n, c, s, y.data_ptr(), _stream()), "zsv_meanpool_fwd")
        ctx.shape = tuple(x.shape)
        ctx.dims = (n, c, s)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        n, c, s = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        with torch.cuda.device(dy.device):
            _lib.check(_lib.load().zsv_meanpool_bwd(dy.data_ptr(), n, c, s, dx.data_ptr(), _stream()), "zsv_meanpool_bwd")
        return dx


def mean_pool(x):
    return _MeanPool.apply(x)


class _MaxPool3d(Function):
    """nn.MaxPool3d with kernel == stride (network.py:103-118)."""

    @staticmethod
    def forward(ctx, x, kernel, padding):
        _require(x)
        x = x.contiguous()
        n, c, ti, hi, wi = (int(v) for v in x.shape)
        kt, kh, kw = kernel
        pt, ph, pw = padding
        to, ho, wo = (ti + 2 * pt - kt) // kt + 1, (hi + 2 * ph - kh) // kh + 1, (wi + 2 * pw - kw) // kw + 1
        y = torch.empty((n, c, to, ho, wo), dtype=torch.float32, device=x.device)
        arg = torch.empty((n, c, to, ho, wo), dtype=torch.int32, device=x.device)
        geom = (n, c, ti, hi, wi, kt, kh, kw, pt, ph, pw, to, ho, wo)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().zsv_maxpool3d_fwd(x.data_ptr(), *geom, y.data_ptr(), arg.data_ptr(), _stream()),
                       "zsv_maxpool3d_fwd")
        ctx.geom = geom
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        geom = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty(geom[:5], dtype=torch.float32, device=dy.device)
        with torch.cuda.device(dy.device):
            _lib.check(_lib.load().zsv_maxpool3d_bwd(dy.data_ptr(), arg.data_ptr(), *geom, dx.data_ptr(), _stream()),
                       "zsv_maxpool3d_bwd")
        return dx, None, None


def max_pool3d(x, kernel_size, stride=None, padding=0):
    k = _triple(kernel_size)
    s = k if stride is None else _triple(stride)
    if s != k:
        raise RuntimeError("max_pool3d: only kernel_size == stride is implemented (all the reference uses)")
    return _MaxPool3d.apply(x, k, _triple(padding))


# ------------------------------------------------------------------------------------------
class _Linear(Function):
    """nn.Linear (network.py:611-616 MLP; :120,132 fc6 / regressor) on the MFMA GEMM core."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        _require(x, weight, bias)
        if x.dim() != 2:
            raise RuntimeError("linear expects a (rows, in_features) input")
        x, weight = x.contiguous(), weight.contiguous()
        rows, fin = int(x.shape[0]), int(x.shape[1])
        fout = int(weight.shape[0])
        if int(weight.shape[1]) != fin:
            raise RuntimeError("linear: in_features mismatch")
        y = torch.empty((rows, fout), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        nbytes = lib.zsv_linear_fwd_workspace_bytes(rows, fin, fout)
        ws = _workspace(nbytes, x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.zsv_linear_fwd(x.data_ptr(), weight.data_ptr(), _ptr(bias), y.data_ptr(), rows, fin,
                                          fout, 1 if relu else 0, _ptr(ws), nbytes, _stream()), "zsv_linear_fwd")
        ctx.dims = (rows, fin, fout)
        ctx.relu = bool(relu)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        rows, fin, fout = ctx.dims
        lib = _lib.load()
        dy = dy.contiguous()
        dx = dw = db = None
        with torch.cuda.device(dy.device):
            if ctx.relu:
                g = torch.empty_like(dy)
                if ctx.has_bias and ctx.needs_input_grad[2]:
                    # ReLU mask and bias gradient in one pass over dy (C3D: every convolution has both)
                    n_, c_ = int(dy.shape[0]), int(dy.shape[1])
                    s_ = dy.numel() // (n_ * c_)
                    db = torch.empty(c_, dtype=torch.float32, device=dy.device)
                    nb = lib.zsv_channel_sum_workspace_bytes(n_, c_, s_)
                    wsb = _workspace(nb, dy.device)
                    _lib.check(lib.zsv_relu_bwd_bias(dy.data_ptr(), y.data_ptr(), g.data_ptr(), n_, c_, s_, db.data_ptr(),
                                                     _ptr(wsb), nb, _stream()), "zsv_relu_bwd_bias")
                else:
                    _lib.check(lib.zsv_relu_bwd(dy.data_ptr(), y.data_ptr(), g.data_ptr(), g.numel(), _stream()), "zsv_relu_bwd")
                dy = g
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                nbytes = lib.zsv_linear_dgrad_workspace_bytes(rows, fin, fout)
                ws = _workspace(nbytes, dy.device)
                _lib.check(lib.zsv_linear_dgrad(dy.data_ptr(), weight.data_ptr(), dx.data_ptr(), rows, fin, fout,
                                                _ptr(ws), nbytes, _stream()), "zsv_linear_dgrad")
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                nbytes = lib.zsv_linear_wgrad_workspace_bytes(rows, fin, fout)
                ws = _workspace(nbytes, dy.device)
                _lib.check(lib.zsv_linear_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), rows, fin, fout, _ptr(ws),
                                                nbytes, _stream()), "zsv_linear_wgrad")
            if ctx.has_bias and ctx.needs_input_grad[2] and db is None:
                db = channel_sum(dy)
        return dx, dw, db, None


def linear(x, weight, bias=None, relu=False):
    return _Linear.apply(x, weight, bias, bool(relu))


def adam_step_(p: torch.Tensor, g: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int,
               lr: float, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """In-place torch.optim.Adam update (main.py:131) of a flat fp32 buffer."""
    _require(p, g, exp_avg, exp_avg_sq)
    for t in (p, g, exp_avg, exp_avg_sq):
        if not t.is_contiguous() or t.numel() != p.numel():
            raise RuntimeError("adam_step_: flat contiguous buffers of equal size expected")
    with torch.cuda.device(p.device):
        _lib.check(_lib.load().zsv_adam_step(p.data_ptr(), g.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                             p.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                             int(step), _stream()), "zsv_adam_step")
