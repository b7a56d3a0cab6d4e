"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` (main.py:61-63,126): instead of
replicating 147 MB of parameters from GPU 0 every forward and reduce-adding gradients back
onto GPU 0, every rank owns a replica, runs its own 22-clip shard (``opt.bs`` is per GPU,
main.py:38,61-63) and the only exchange per step is one averaged all-reduce of the live
gradients (126.9 MB fp32), issued bucket by bucket while backward is still running:

* buckets are filled in the order gradients are actually produced (recorded during the first
  step): MLP head -> layer4 -> ... -> stem, so 74 % of the bytes (layer4) are on the wire
  after ~6 % of the backward FLOPs;
* each bucket is a flat fp32 buffer; when its last gradient lands, an event recorded on the
  autograd stream gates a side HIP stream that packs the bucket, runs ``all_reduce(SUM)``
  (backend "nccl" == RCCL on ROCm), scales by 1/world and unpacks -- backward never waits;
* ``finish_step()`` makes the compute stream wait for the side stream before
  ``optimizer.step()``;
* parameters that never receive a gradient (the reference's unused Transformer encoder,
  ``model.fc``, ... -- SURVEY F5) are discovered in the first step and excluded, so nothing
  waits for them;
* BatchNorm statistics stay per replica, as under DataParallel.

After ``finish_step()`` every live parameter's ``.grad`` IS a view of its bucket (the averaged values are not
copied back), so ``optim.FusedAdam(..., grad_buckets=sync)`` walks the all-reduce buffers themselves with a
descriptor table built once, and ``torch.optim.Adam`` sees ordinary ``.grad`` tensors.
``GradientSync(model, local=True)`` keeps the bucketing without any collective (one process, one GPU).

With equal shards, the mean over ranks of per-rank mean-MSE gradients equals the full-batch
mean-MSE gradient DataParallel computes.  Works on CPU tensors with the gloo backend (tests).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import weakref

import torch
import torch.distributed as dist

from . import _lib

DEFAULT_BUCKET_BYTES = 25 * 1024 * 1024


@torch.no_grad()
def broadcast_tensors(tensors, src: int = 0, group=None) -> int:
    """Every rank takes rank ``src``'s values of ``tensors`` (parameters, buffers): ONE collective per dtype over a flat
    staging buffer instead of one per tensor (R(2+1)D-18: 304 tensors -> 2 broadcasts).  Returns the number of collectives.
    The source rank's tensors are not written."""
    by_dtype: Dict[torch.dtype, List[torch.Tensor]] = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t.data)
    rank = dist.get_rank()
    calls = 0
    for dtype, group_tensors in by_dtype.items():
        flat = torch.cat([t.reshape(-1) for t in group_tensors])
        dist.broadcast(flat, src=src, group=group)
        calls += 1
        if rank != src:
            off = 0
            for t in group_tensors:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
    return calls


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        p0 = params[0]
        self.flat = torch.empty(self.numel, dtype=p0.dtype, device=p0.device)
        self.views = []
        off = 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.pending = len(params)
        self.work = None


class GradientSync:
    """Bucketed, backward-overlapped gradient averaging for one model replica."""

    def __init__(self, model: torch.nn.Module, process_group=None, bucket_bytes: int = DEFAULT_BUCKET_BYTES,
                 broadcast_initial_state: bool = True, local: bool = False):
        self.local = bool(local)
        if not self.local and not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (one process per GPU; init_process_group first)")
        self.group = process_group
        self.world = 1 if self.local else dist.get_world_size(process_group)
        self.layout_version = 0                  # bumped whenever the buckets are (re)built
        self.model = model
        self.bucket_bytes = int(bucket_bytes)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._index: Dict[int, int] = {id(p): i for i, p in enumerate(self.params)}
        self._arrival: List[int] = []          # parameter indices in the order their grads landed
        self._buckets: Optional[List[_Bucket]] = None
        self._bucket_of: Dict[int, _Bucket] = {}
        self._in_step = False
        self._cuda = any(p.is_cuda for p in self.params)
        self._side = torch.cuda.Stream() if self._cuda else None
        # (ops._dw_read_early: a post-accumulate hook normally reads `.grad` on the backward stream, so the weight-gradient side
        # stream is joined before it runs; THIS hook joins that stream itself in `_launch`, and says so to keep the overlap)
        owner = weakref.ref(self)

        def on_grad(p):
            me = owner()
            if me is not None:
                me._on_grad(p)
        on_grad._zsv_joins_wgrad = True
        self._hooks = [p.register_post_accumulate_grad_hook(on_grad) for p in self.params]
        self.bytes_reduced_last_step = 0
        if broadcast_initial_state and self.world > 1:
            self.broadcast_state()

    # -- setup ---------------------------------------------------------------------------
    @torch.no_grad()
    def broadcast_state(self, src: int = 0) -> None:
        """Every replica starts from rank ``src``'s parameters and buffers (``src``: a global rank)."""
        broadcast_tensors(list(self.model.parameters()) + list(self.model.buffers()), src, self.group)
        if dist.get_rank() != src:
            _lib.note_raw_write()                  # `.data` writes do not bump the version counters

    def _build_buckets(self, order: List[int]) -> None:
        buckets, cur, cur_bytes = [], [], 0
        for idx in order:
            p = self.params[idx]
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > self.bucket_bytes:
                buckets.append(_Bucket(cur))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            buckets.append(_Bucket(cur))
        self._buckets = buckets
        self._bucket_of = {id(p): b for b in buckets for p in b.params}
        self.layout_version += 1

    @property
    def ready(self) -> bool:
        """True once the discovery step has fixed the live set and the bucket layout."""
        return self._buckets is not None

    def bucket_layout(self):
        """``[(flat buffer, [(parameter, element offset), ...]), ...]`` in production order."""
        out = []
        for b in self._buckets or []:
            off, rows = 0, []
            for p in b.params:
                rows.append((p, off))
                off += p.numel()
            out.append((b.flat, rows))
        return out

    @property
    def live_parameter_count(self) -> int:
        return 0 if self._buckets is None else sum(len(b.params) for b in self._buckets)

    @property
    def bucket_sizes(self) -> List[int]:
        return [] if self._buckets is None else [b.numel for b in self._buckets]

    # -- per step ------------------------------------------------------------------------
    def begin_step(self) -> None:
        self._in_step = True
        self._arrival = []
        self.bytes_reduced_last_step = 0
        if self._buckets is not None:
            for b in self._buckets:
                b.pending = len(b.params)
                b.work = None

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if not self._in_step:
            return
        if self._buckets is None:                  # discovery step: just record the order
            self._arrival.append(self._index[id(p)])
            return
        b = self._bucket_of.get(id(p))
        if b is None:
            raise RuntimeError("a parameter that produced no gradient in the first step produced one now; "
                               "rebuild GradientSync (the live set is fixed after discovery)")
        if b.pending <= 0:
            raise RuntimeError("a gradient arrived for a bucket that was already reduced this step: two backward "
                               "passes per begin_step() (gradient accumulation) are not supported")
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    @torch.no_grad()
    def _launch(self, b: _Bucket) -> None:
        """Pack the bucket's gradients into its flat buffer, sum over ranks, scale by 1/world -- on the side
        stream.  Nothing is copied back: ``finish_step`` points each ``.grad`` at its slice of the buffer."""
        grads = [p.grad for p in b.params]
        scale = 1.0 / self.world
        if self._cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            self._side.wait_event(ready)
            from . import ops
            ops.join_wgrad_streams(self._side)        # weight gradients may come from their own side stream
            with torch.cuda.stream(self._side):
                torch._foreach_copy_(b.views, grads)
                if self.world > 1:
                    b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                    b.work.wait()                 # orders the side stream after RCCL; host does not block
                    b.flat.mul_(scale)
                for g in grads:                   # the side stream reads memory owned by the autograd stream
                    g.record_stream(self._side)
        else:
            torch._foreach_copy_(b.views, grads)
            if self.world > 1:
                dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group)
                b.flat.mul_(scale)
        self.bytes_reduced_last_step += b.numel * b.flat.element_size()

    def finish_step(self) -> None:
        """Call after ``loss.backward()`` and before ``optimizer.step()``."""
        if not self._in_step:
            raise RuntimeError("finish_step() without begin_step()")
        self._in_step = False
        if self._buckets is None:
            # first step: the live set and the production order are now known.  Ranks must agree
            # on them (same model, same graph) -- checked cheaply through the count.
            order = list(self._arrival)
            if self.local:
                self._build_buckets(order)
                for b in self._buckets:
                    self._launch(b)
                self._adopt_bucket_views()
                return
            count = torch.tensor([len(order)], dtype=torch.int64, device=self.params[0].device)
            lo, hi = count.clone(), count.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if lo.item() != hi.item():
                raise RuntimeError("ranks disagree on which parameters receive gradients")
            # every rank lays its buckets out in RANK 0's production order: a flat all-reduce of buckets built
            # from different local orders would silently add unrelated gradients.  The live SET must agree.
            mine = torch.tensor(order, dtype=torch.int64, device=count.device)
            ref = mine.clone()
            src = 0 if self.group is None else dist.get_global_rank(self.group, 0)
            dist.broadcast(ref, src=src, group=self.group)
            differs = torch.tensor([0 if torch.equal(torch.sort(ref)[0], torch.sort(mine)[0]) else 1],
                                   dtype=torch.int64, device=count.device)
            dist.all_reduce(differs, op=dist.ReduceOp.MAX, group=self.group)
            if differs.item():
                raise RuntimeError("ranks disagree on which parameters receive gradients (same count, different set)")
            order = [int(i) for i in ref.tolist()]
            self._build_buckets(order)
            for b in self._buckets:                # no overlap in the discovery step
                self._launch(b)
        else:
            late = [b for b in self._buckets if b.pending != 0]
            if late:
                missing = sum(b.pending for b in late)
                raise RuntimeError(f"{missing} live parameters produced no gradient this step")
        self._adopt_bucket_views()

    def _adopt_bucket_views(self) -> None:
        if self._cuda:
            torch.cuda.current_stream().wait_stream(self._side)
        for b in self._buckets:
            for p, v in zip(b.params, b.views):
                if p.grad is not v:
                    p.grad = v                    # the averaged gradient lives in the bucket; no copy back

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
