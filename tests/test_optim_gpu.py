"""Optimizer step of main.py:131-137,195-203 on the device: FusedAdam over the flat gradient buckets and the
device-resident loss scaler, against ``torch.optim.Adam`` + ``torch.amp.GradScaler``."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from zeroshotvideoclassification_amd import ddp, optim  # noqa: E402

DEV = "cuda"


def _params(seed, device):
    g = torch.Generator().manual_seed(seed)
    shapes = [(5000,), (3, 7, 11), (1,), (4097,), (64, 64), (300, 512)]
    return [torch.randn(*s, generator=g).to(device).requires_grad_() for s in shapes]


def _loss(params, step, poison):
    """sum_i <p_i, w_i(step)>: the gradient of p_i is w_i.  ``poison`` puts an inf / nan into one w."""
    g = torch.Generator().manual_seed(1000 + step)
    total = 0.0
    for i, p in enumerate(params):
        w = torch.randn(p.shape, generator=g)
        if poison is not None and i == 3:
            w.view(-1)[17] = poison
        total = total + (p * w.to(p.device)).sum()
    return total


def test_loss_scaler_matches_torch_gradscaler_with_injected_infs():
    """scaler.scale(loss).backward(); scaler.step(opt); scaler.update() for 12 steps with inf / nan gradients at
    steps 2, 5 and 6: skipped updates, scale back-off and growth (interval 3), the Adam bias correction counting
    only the steps actually taken -- parameters after every step, scale and step count equal torch's."""
    ref_p, dev_p = _params(4, DEV), _params(4, DEV)
    ref_opt = torch.optim.Adam(ref_p, lr=1e-2)
    ref_scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    dev_opt = optim.FusedAdam(dev_p, lr=1e-2)
    dev_scaler = optim.LossScaler(init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    poison = {2: float("inf"), 5: float("nan"), 6: float("-inf")}
    for step in range(12):
        ref_opt.zero_grad(set_to_none=True)
        ref_scaler.scale(_loss(ref_p, step, poison.get(step))).backward()
        ref_scaler.step(ref_opt)
        ref_scaler.update()
        dev_opt.zero_grad(set_to_none=True)
        dev_scaler.scale(_loss(dev_p, step, poison.get(step))).backward()
        dev_scaler.step(dev_opt)
        dev_scaler.update()
        st = dev_scaler.state()
        assert st["scale"] == ref_scaler.get_scale(), (step, st)
        assert st["found_inf"] == 0
        for i, (a, b) in enumerate(zip(dev_p, ref_p)):
            err = (a.detach() - b.detach()).abs().max().item()
            assert err <= 2e-6 * b.detach().abs().max().item(), (step, i, err)
    st = dev_scaler.state()
    assert st["steps_done"] == 12 - len(poison)
    assert float(ref_opt.state[ref_p[0]]["step"]) == st["steps_done"]
    assert float(dev_opt.state_dict()["state"][0]["step"]) == st["steps_done"]
    with pytest.raises(RuntimeError, match="LossScaler"):
        dev_opt.step()                                                       # a scaled optimizer is stepped through its scaler
    with pytest.raises(RuntimeError, match="FusedAdam"):
        dev_scaler.step(ref_opt)


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(40, 300)
        self.b = torch.nn.Linear(300, 300)
        self.c = torch.nn.Linear(300, 20)
        self.dead = torch.nn.Linear(4, 4)                                    # never used: no gradient, no bucket

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


@pytest.mark.parametrize("use_scaler", [False, True])
def test_fused_adam_walks_the_gradient_buckets(use_scaler):
    """GradientSync(local=True) packs the gradients into flat buckets and leaves ``.grad`` as views of them;
    FusedAdam(grad_buckets=...) then steps from a descriptor table built once (moments in matching flat buffers).
    Same trajectory as torch.optim.Adam on a copy, through the discovery step (dynamic table) and the static path."""
    torch.manual_seed(3)
    model = _Net().to(DEV)
    ref = copy.deepcopy(model)
    sync = ddp.GradientSync(model, bucket_bytes=200 * 1024, local=True)
    opt = optim.FusedAdam(model.parameters(), lr=1e-2, grad_buckets=sync)
    ref_opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    scaler = optim.LossScaler(init_scale=256.0) if use_scaler else None
    g = torch.Generator().manual_seed(9)
    from zeroshotvideoclassification_amd import train
    crit = torch.nn.MSELoss()
    for step in range(5):
        x, z = torch.randn(16, 40, generator=g).to(DEV), torch.randn(16, 20, generator=g).to(DEV)
        train.train_step(model, opt, crit, x, z, sync, scaler)
        ref_opt.zero_grad(set_to_none=True)
        crit(ref(x), z).backward()
        ref_opt.step()
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            assert (p - q).abs().max().item() <= 1e-5 * (q.abs().max().item() + 1e-6), (step, k)
        if step >= 1:
            assert opt._static is not None and opt._static[0] == sync.layout_version
    assert len(sync.bucket_sizes) >= 2 and sync.live_parameter_count == 6
    # the gradients ARE slices of the flat buffers, the moments are slices of their flat twins
    for flat, rows in sync.bucket_layout():
        for p, off in rows:
            assert p.grad.data_ptr() == flat.data_ptr() + 4 * off
    m_flat, _ = opt._static[5][0]
    first = sync.bucket_layout()[0][1][0][0]
    assert opt.state[first]["exp_avg"].data_ptr() == m_flat.data_ptr()
    assert model.dead.weight.grad is None and not opt.state[model.dead.weight]
    # a checkpoint round trip keeps working (moments re-homed into fresh flat buffers)
    sd = copy.deepcopy(opt.state_dict())
    if not use_scaler:
        opt2 = optim.FusedAdam(model.parameters(), lr=1e-2, grad_buckets=sync)
        opt2.load_state_dict(sd)
        x, z = torch.randn(16, 40, generator=g).to(DEV), torch.randn(16, 20, generator=g).to(DEV)
        train.train_step(model, opt2, crit, x, z, sync)
        ref_opt.zero_grad(set_to_none=True)
        crit(ref(x), z).backward()
        ref_opt.step()
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            assert (p - q).abs().max().item() <= 1e-5 * (q.abs().max().item() + 1e-6), ("reloaded", k)


def test_inf_in_a_later_group_skips_every_group():
    """GradScaler.step checks all gradients before the optimizer updates anything: with two parameter groups and a
    non-finite gradient only in the SECOND, the first group must stay untouched as well (then both step normally)."""
    pa, pb = _params(11, DEV), _params(12, DEV)
    before = [p.detach().clone() for p in pa + pb]
    opt = optim.FusedAdam([{"params": pa}, {"params": pb, "lr": 3e-3}], lr=1e-2)
    scaler = optim.LossScaler(init_scale=64.0)
    ref_a, ref_b = [p.detach().clone().requires_grad_() for p in pa], [p.detach().clone().requires_grad_() for p in pb]
    ref_opt = torch.optim.Adam([{"params": ref_a}, {"params": ref_b, "lr": 3e-3}], lr=1e-2)
    ref_scaler = torch.amp.GradScaler("cuda", init_scale=64.0)
    for step, poison in enumerate([float("inf"), None, None]):
        opt.zero_grad(set_to_none=True)
        scaler.scale(_loss(pa, step, None) + _loss(pb, 50 + step, poison)).backward()
        scaler.step(opt)
        scaler.update()
        ref_opt.zero_grad(set_to_none=True)
        ref_scaler.scale(_loss(ref_a, step, None) + _loss(ref_b, 50 + step, poison)).backward()
        ref_scaler.step(ref_opt)
        ref_scaler.update()
        if step == 0:
            for p, q in zip(pa + pb, before):
                assert torch.equal(p.detach(), q)                    # nothing moved, in either group
        for a, b in zip(pa + pb, ref_a + ref_b):
            assert (a.detach() - b.detach()).abs().max().item() <= 2e-6 * b.detach().abs().max().item()
    assert scaler.state()["steps_done"] == 2 and scaler.get_scale() == ref_scaler.get_scale()


def test_resume_under_the_loss_scaler():
    """Checkpoint / resume of the scaled optimizer: LossScaler.state_dict() / load_state_dict() carry scale, growth tracker
    and the count of steps taken; a FusedAdam restored with load_state_dict() accepts the scaler afterwards and continues the
    trajectory of an uninterrupted run."""
    def run(split_at):
        ps = _params(21, DEV)
        opt = optim.FusedAdam(ps, lr=1e-2)
        scaler = optim.LossScaler(init_scale=512.0, growth_interval=2)
        for step in range(6):
            if step == split_at:
                sd_o, sd_s = copy.deepcopy(opt.state_dict()), scaler.state_dict()
                opt = optim.FusedAdam(ps, lr=1e-2)
                opt.load_state_dict(sd_o)
                scaler = optim.LossScaler(init_scale=1.0)
                scaler.load_state_dict(sd_s)
            opt.zero_grad(set_to_none=True)
            scaler.scale(_loss(ps, step, float("nan") if step == 1 else None)).backward()
            scaler.step(opt)
            scaler.update()
        return [p.detach().clone() for p in ps], scaler.state_dict()

    straight, s0 = run(None)
    resumed, s1 = run(3)
    assert s0 == s1 and s0["steps_done"] == 5 and set(s0) >= {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
    for a, b in zip(straight, resumed):
        assert torch.equal(a, b)
    # a torch GradScaler state dict (no steps_done) is accepted: the optimizer's loaded step count seeds the scaler
    ps = _params(21, DEV)
    opt = optim.FusedAdam(ps, lr=1e-2)
    for step in range(2):
        opt.zero_grad(set_to_none=True)
        _loss(ps, step, None).backward()
        opt.step()
    opt2 = optim.FusedAdam(ps, lr=1e-2)
    opt2.load_state_dict(copy.deepcopy(opt.state_dict()))
    scaler = optim.LossScaler()
    scaler.load_state_dict(torch.amp.GradScaler("cuda", init_scale=128.0).state_dict())
    opt2.zero_grad(set_to_none=True)
    scaler.scale(_loss(ps, 2, None)).backward()
    scaler.step(opt2)
    scaler.update()
    assert scaler.state()["steps_done"] == 3 and scaler.get_scale() == 128.0
