"""Training / evaluation step of the reference's driver, on the HIP path.

``train_step`` reproduces the order of ``train_one_epoch`` (main.py:170-203): zero_grad ->
forward -> MSE -> backward -> optimizer step, with ``main_02.py:256``'s handling of the
``(emb, None)`` tuple.  The reference wraps forward in CUDA fp16 autocast +
GradScaler (main.py:137,172,195-203); the fp32 configuration benchmarked here (BASELINE
configs 1-3) runs without them, which is what the reference's CPU path does too (autocast is a
no-op there).  The per-step train accuracy (main.py:182-185) is computed on the device so it
does not stall the queue.

``evaluate`` / ``compute_accuracy`` restate main.py:224-325: eval-mode forward under
``no_grad``, cosine nearest class in the 300-d embedding space, top-1 / top-5, and the ten
seeded half-class splits.

``load_weights`` / ``save_checkpoint`` keep the reference's checkpoint format (main.py:114-124,
361-365): ``{'state_dict' ('module.'-prefixed keys), 'opt', 'accuracy'}``, loaded by key
intersection, so checkpoints move between the two implementations unchanged.
"""
from __future__ import annotations

from typing import Iterable, Optional, Sequence, Tuple

from ctypes import c_void_p

import numpy as np
import torch

from . import _lib, ops


_PREFIX = "module."       # nn.DataParallel's prefix in saved checkpoints (main.py:116,126,363)


def load_weights(model: torch.nn.Module, path: str) -> int:
    """main.py:114-124: strip the ``module.`` prefix, keep the keys the model has, load the rest
    from the model itself.  Returns the number of tensors taken from the checkpoint."""
    weights = torch.load(path, map_location="cpu", weights_only=False)["state_dict"]
    own = getattr(model, "module", model)
    model_dict = own.state_dict()
    j = len(_PREFIX)
    weights = {k[j:]: v for k, v in weights.items() if k[j:] in model_dict}
    model_dict.update(weights)
    own.load_state_dict(model_dict)
    _lib.note_raw_write()
    return len(weights)


def save_checkpoint(model: torch.nn.Module, path: str, opt=None, accuracy: float = 0.0) -> None:
    """main.py:361-365: the saved keys carry the ``module.`` prefix whether or not the model is wrapped."""
    own = getattr(model, "module", model)
    state = {_PREFIX + k: v.detach().cpu() for k, v in own.state_dict().items()}
    torch.save({"state_dict": state, "opt": opt, "accuracy": accuracy}, path)


def embed(model: torch.nn.Module, x: torch.Tensor) -> torch.Tensor:
    out = model(x)
    return out[0] if isinstance(out, tuple) else out


class StepPacer:
    """Bounds how far the host may run ahead of the device: ``wait()`` at the head of a step blocks until the step
    ``depth`` steps back has finished on the device.

    The reference's loop reads ``loss.item()`` every iteration (main.py:203), i.e. it never runs ahead at all.  This path
    has no host sync in a step, and a host that queues step k+1 while step k still runs cannot reuse the activation blocks
    the weight-gradient stream still holds (``Tensor.record_stream``): every step of lead costs one more set of saved
    activations (8-11 GB at 22 clips) taken from the driver by hipMalloc inside the loop.  With few launches per step
    (C3D: ~130) the host would be many steps ahead within milliseconds.  ``depth = 2`` keeps one whole step queued behind
    the running one (the device never idles) and the pool at three sets, reached after three steps."""

    def __init__(self, depth: int = 2):
        from collections import deque
        self.depth = max(1, int(depth))
        self.marks = deque()
        self.waits = 0                 # steps whose head really had to wait (diagnostics)

    def wait(self) -> None:
        while len(self.marks) >= self.depth:
            ev = self.marks.popleft()
            if not ev.query():
                self.waits += 1
                ev.synchronize()

    def mark(self) -> None:
        ev = torch.cuda.Event()
        ev.record()
        self.marks.append(ev)

    def drain(self) -> None:
        self.marks.clear()


def train_step(model: torch.nn.Module, optimizer: torch.optim.Optimizer, criterion, x: torch.Tensor,
               z: torch.Tensor, grad_sync=None, scaler=None, pacer: Optional[StepPacer] = None,
               autocast: bool = False, graph: Optional[bool] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """One iteration of main.py:170-203.  ``grad_sync`` (a ``ddp.GradientSync``) all-reduces the
    gradients across ranks, overlapped with backward, before the optimizer step.  ``scaler`` (an
    ``optim.LossScaler`` or a ``torch.cuda.amp.GradScaler``) reproduces main.py:195-203:
    ``scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()`` -- a non-finite gradient skips
    the update and halves the scale.  ``autocast=True`` runs forward and loss inside ``amp.autocast()`` -- main.py:172's
    ``with autocast():`` -- i.e. the VideoResNet trunks in bf16 (``amp``: bf16 activations and products, fp32 accumulation,
    statistics, parameters, gradients and loss); the default is the fp32 step of BASELINE configs 1-3, which is what the
    reference's CPU path runs (autocast is a no-op there).  ``pacer`` (a ``StepPacer``) bounds the host's lead over the device."""
    if pacer is not None:
        pacer.wait()
    optimizer.zero_grad(set_to_none=True)
    if grad_sync is not None:
        grad_sync.begin_step()
    if autocast:
        from . import amp
        with amp.autocast(graph=graph):           # graph=True: the bf16 trunk as two hipGraphs (amp._GraphedTrunk)
            y = embed(model, x)
            loss = criterion(y, z)
    else:
        y = embed(model, x)
        loss = criterion(y, z)
    (scaler.scale(loss) if scaler is not None else loss).backward()
    ops.join_wgrad_streams()          # (the autograd end-of-pass callback has done this already; a no-op wait then)
    if grad_sync is not None:
        grad_sync.finish_step()
    if scaler is not None:
        scaler.step(optimizer)
        scaler.update()
    else:
        optimizer.step()
    if pacer is not None:
        pacer.mark()
    return y.detach(), loss.detach()


def nearest_classes(embed_: torch.Tensor, class_embed: torch.Tensor, k: int = 5) -> torch.Tensor:
    """``cdist(embed, class_embed, 'cosine').argsort(1)[:, :k]`` (main.py:321) on the device: double-precision
    cosine distances on the matrix core + per-row top-k with ties by lower index (``zsv_cosine_topk``,
    csrc/nearest_class.hip).  ``(rows, k)`` int64 class indices; HIP tensors only."""
    if not (embed_.is_cuda and class_embed.is_cuda):
        raise RuntimeError("nearest_classes runs on an MI355X HIP device only (no CPU fallback)")
    c = class_embed.detach().float().contiguous()
    e = embed_.detach().float().reshape(len(embed_), c.shape[-1] if c.dim() == 2 else -1).contiguous()
    if e.dim() != 2 or c.dim() != 2 or e.shape[1] != c.shape[1]:
        raise RuntimeError(f"nearest_classes: embeddings {tuple(e.shape)} vs class table {tuple(c.shape)}")
    rows, n_classes = int(e.shape[0]), int(c.shape[0])
    k = min(int(k), n_classes)
    out = torch.empty((rows, k), dtype=torch.int32, device=e.device)
    if rows == 0:
        return out.long()
    lib = _lib.load()
    nbytes = lib.zsv_cosine_topk_workspace_bytes(rows, n_classes)
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=e.device)
    with torch.cuda.device(e.device):
        _lib.check(lib.zsv_cosine_topk(e.data_ptr(), c.data_ptr(), rows, int(e.shape[1]), n_classes, k, out.data_ptr(),
                                       None, ws.data_ptr(), nbytes, c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "zsv_cosine_topk")
    return out.long()


def cosine_ranking(pred: torch.Tensor, class_embed: torch.Tensor, k: int = 5) -> torch.Tensor:
    """The first ``k`` columns of the reference's ``cdist(...).argsort(1)`` (main.py:321)."""
    return nearest_classes(pred, class_embed, k)


def train_accuracy(y: torch.Tensor, class_embed: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """main.py:182-185 without leaving the device: percent of samples whose nearest class embedding is
    their label's (``cdist(Y, class_embed, 'cosine').argmin(1) == l``).  A 0-d device tensor, no host sync."""
    pred = nearest_classes(y, class_embed.to(y.device), 1)[:, 0]
    return (pred == labels.to(pred.device).reshape(-1)).float().mean() * 100.0


def compute_accuracy(predicted_embed: torch.Tensor, class_embed: torch.Tensor,
                     true_embed: torch.Tensor) -> Tuple[float, float]:
    """Top-1 / top-5 accuracy (percent) to the closest class embedding (main.py:316-325)."""
    assert len(predicted_embed) == len(true_embed), "True and predicted labels must have the same number of samples"
    order = nearest_classes(predicted_embed, class_embed, 5)
    y = nearest_classes(true_embed, class_embed, 1)[:, 0]
    n = max(len(y), 1)
    # counts are exact integers; the reference's np.mean of booleans is the same count / n in double
    top1 = float((order[:, 0] == y).sum().item()) / n * 100
    top5 = float((order == y[:, None]).any(dim=1).sum().item()) / n * 100
    return top1, top5


def _gather_rows(t: torch.Tensor, group) -> torch.Tensor:
    """All ranks' rows of ``t`` (different counts per rank), concatenated in rank order."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    count = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    width = max(counts)
    padded = torch.zeros((width,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    padded[:t.shape[0]] = t
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)])


@torch.no_grad()
def evaluate(model: torch.nn.Module, batches: Iterable[Sequence[torch.Tensor]], class_embed: torch.Tensor,
             device: Optional[torch.device] = None, splits: int = 10, dtype: Optional[torch.dtype] = None,
             group=None, sharded: Optional[bool] = None, local_batches: bool = False,
             sync_state: bool = True) -> dict:
    """main.py:224-313 for one test set.  ``batches`` yields ``(X, labels, Z, ...)``; samples with
    label -1 (failed loads, auxiliary_dataset.py:502-505) are dropped like main.py:246-248.
    ``dtype=torch.bfloat16`` runs the forward on the bf16 engine (``inference.Bf16Engine``, the
    reduced-precision eval of BASELINE config 5 / the reference's autocast, main.py:172), ``torch.float32``
    on the folded fp32 engine (``inference.Fp32Engine``); default (None): the module's own fp32 forward.

    With ``torch.distributed`` initialised (one process per GPU; ``sharded`` defaults to that) every rank
    iterates the SAME ``batches`` and runs the forward for every ``world``-th one (batch ``i`` belongs to
    rank ``i % world``); with ``local_batches=True`` the iterable is this rank's OWN shard (a per-rank loader:
    nothing is skipped, no rank decodes another rank's clips).  The ``(pred, true, label)`` rows are
    all-gathered over ``group`` (RCCL) and every rank computes the same accuracies.  The reference evaluates
    under one-process ``nn.DataParallel`` (main.py:126,250), which scatters each batch and runs every replica
    with DEVICE 0's parameters and BatchNorm running statistics.  Data-parallel training keeps those statistics
    per replica (``ddp.GradientSync``), so before a sharded evaluation every rank takes rank 0's parameters and
    buffers (``sync_state``; one flat broadcast per dtype): all samples are scored by ONE model -- the one rank 0
    would checkpoint (main.py:361-365).  NOTE: this overwrites the other ranks' BatchNorm running statistics, mid-training
    too -- what ``nn.DataParallel`` does implicitly every forward.  ``sync_state=False`` skips the broadcast when the caller knows the
    replicas are identical (e.g. right after ``load_weights`` on every rank)."""
    import torch.distributed as dist
    if sharded is None:
        sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if sharded else (0, 1)
    was_training = model.training
    model.eval()
    if sharded and sync_state:
        src = 0 if group is None else dist.get_global_rank(group, 0)
        from .ddp import broadcast_tensors
        broadcast_tensors(list(model.parameters()) + list(model.buffers()), src, group)      # one collective per dtype
        if dist.get_rank() != src:                         # (the source's values did not change: its panels / engines stay)
            _lib.note_raw_write()                          # `.data` writes: cached folded-BatchNorm engines must rebuild
    device = device or next(model.parameters()).device
    forward = model
    if dtype in (torch.bfloat16, torch.float32):
        from .inference import engine_for
        forward = engine_for(model, dtype)                 # BatchNorm folded; fp32 or bf16 activations
    elif dtype is not None:
        raise RuntimeError(f"evaluate: dtype {dtype} is not supported (fp32 or bf16)")
    preds, trues, labels = [], [], []
    for index, batch in enumerate(batches):
        if not local_batches and index % world != rank:
            continue
        x, l, z = batch[0], batch[1], batch[2]
        keep = l != -1
        if keep.sum() == 0:
            continue
        x, l, z = x[keep], l[keep], z[keep]
        preds.append(embed(forward, x.to(device, non_blocking=True)).float())
        # targets / labels join the device once, after the loop: a per-batch blocking copy would drain the
        # queue every batch (and an idle-then-busy GPU showed sporadic 30-80 ms stalls on the test pool)
        trues.append(z.float().reshape(len(l), -1))
        labels.append(l.reshape(-1))
    model.train(was_training)
    class_embed = class_embed.to(device)
    width = int(class_embed.shape[1])
    pred = torch.cat(preds) if preds else torch.zeros((0, width), device=device)
    true = torch.cat(trues).to(device) if trues else torch.zeros((0, width), device=device)
    label_dev = torch.cat(labels).to(device).long() if labels else torch.zeros((0,), dtype=torch.int64, device=device)
    if sharded:
        pred, true, label_dev = (_gather_rows(t, group) for t in (pred, true, label_dev))
    label = label_dev.cpu().numpy()
    acc, acc5 = compute_accuracy(pred, class_embed, true)
    out = {"accuracy": acc, "accuracy_top5": acc5, "n": int(len(pred))}
    if splits:
        a1, a5 = [], []
        for split in range(splits):
            np.random.seed(split)                                     # main.py:284
            sel_classes = np.random.permutation(len(class_embed))[:len(class_embed) // 2]
            sel = torch.from_numpy(np.isin(label, sel_classes)).to(device)
            if sel.sum() == 0:
                continue
            s1, s5 = compute_accuracy(pred[sel], class_embed[torch.from_numpy(sel_classes).to(device)], true[sel])
            a1.append(s1)
            a5.append(s5)
        if a1:
            out.update(split_accuracy=float(np.mean(a1)), split_accuracy_std=float(np.std(a1)),
                       split_accuracy_top5=float(np.mean(a5)), split_accuracy_top5_std=float(np.std(a5)))
    return out
