"""world_size-2 data-parallel gradient exchange on CPU tensors with the gloo backend."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Net(torch.nn.Module):
    """Small stand-in with the features that matter to GradientSync: several parameter tensors of
    different sizes, a frozen one, and one that never receives a gradient."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(16, 32)
        self.b = torch.nn.Linear(32, 32)
        self.c = torch.nn.Linear(32, 8)
        self.dead = torch.nn.Linear(4, 4)          # never used in forward
        self.frozen = torch.nn.Linear(8, 8)
        for p in self.frozen.parameters():
            p.requires_grad = False
        self.register_buffer("running", torch.zeros(3))

    def forward(self, x):
        return self.frozen(self.c(torch.relu(self.b(torch.relu(self.a(x))))))


def _worker(rank, world, port, bucket_bytes, result_dir):
    sys.path.insert(0, ROOT)
    from zeroshotvideoclassification_amd import ddp, train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                      # replicas start DIFFERENT on purpose
        model = _Net()
        sync = ddp.GradientSync(model, bucket_bytes=bucket_bytes)
        # broadcast_state made every replica equal to rank 0
        torch.manual_seed(100)
        ref0 = _Net()
        for (k, p), (_, q) in zip(model.state_dict().items(), ref0.state_dict().items()):
            assert torch.equal(p, q), k

        g = torch.Generator().manual_seed(7)
        full_x = torch.randn(world * 6, 16, generator=g)
        full_z = torch.randn(world * 6, 8, generator=g)
        x, z = full_x[rank * 6:(rank + 1) * 6], full_z[rank * 6:(rank + 1) * 6]

        # single-process full-batch reference (what DataParallel computes)
        torch.manual_seed(100)
        ref = _Net()
        ref_opt = torch.optim.Adam([p for p in ref.parameters() if p.requires_grad], lr=1e-2)
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
        crit = torch.nn.MSELoss()
        for step in range(3):
            train.train_step(model, opt, crit, x, z, sync)
            ref_opt.zero_grad()
            crit(ref(full_x), full_z).backward()
            if step == 0:
                for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
                    if q.grad is None:
                        assert p.grad is None, k
                    else:
                        assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7), k
            ref_opt.step()
            for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
                assert torch.allclose(p, q, rtol=1e-4, atol=1e-6), (step, k)
        assert sync.live_parameter_count == 6              # a, b, c weights+biases; dead/frozen excluded
        assert sum(sync.bucket_sizes) == sum(p.numel() for n, p in model.named_parameters()
                                             if n.split(".")[0] in ("a", "b", "c"))
        assert sync.bytes_reduced_last_step == 4 * sum(sync.bucket_sizes)
        # production order: the head's gradients are ready first (reverse of forward)
        first = sync._buckets[0].params[0]
        assert any(first is p for p in model.c.parameters())
        open(os.path.join(result_dir, f"ok{rank}_{len(sync.bucket_sizes)}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes,min_buckets", [(25 * 1024 * 1024, 1), (1024, 3)])
def test_gradient_sync_equals_full_batch_training(tmp_path, bucket_bytes, min_buckets):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, bucket_bytes, str(tmp_path)), nprocs=world, join=True)
    done = sorted(os.listdir(tmp_path))
    assert [d.split("_")[0] for d in done] == ["ok0", "ok1"]
    assert all(int(d.split("_")[1]) >= min_buckets for d in done)


def test_gradient_sync_requires_process_group():
    sys.path.insert(0, ROOT)
    from zeroshotvideoclassification_amd import ddp
    if dist.is_initialized():
        pytest.skip("a process group is already up")
    with pytest.raises(RuntimeError, match="not initialised"):
        ddp.GradientSync(_Net())
