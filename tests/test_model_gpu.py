"""End-to-end parity of the HIP model path against the reference-generated golden vectors.

north_star tolerance: embeddings and MSE loss within 1e-3 relative (fp32).  Tighter bounds
are asserted where the fp32-vs-fp64 noise floor allows: 1e-4 on embeddings
/ loss / BatchNorm statistics, gradient norms per parameter within 3e-2 of the fp64 oracle
(BN-backward cancellation puts fp32 CPU itself 1e-2 away from fp64).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from helpers import case_inputs, load_golden, make_opt, rel_err, rel_l2, sample_idx  # noqa: E402
from zeroshotvideoclassification_amd import network, synthetic, train  # noqa: E402

DEV = "cuda"
EMB_TOL = 1e-3          # north_star bar
TIGHT = 1e-4


def build(case):
    g = load_golden(case)
    model = network.get_network(make_opt(str(g["meta_network"])))
    weights = synthetic.keyed_state_dict(model.state_dict(), seed=0, bn_jitter=bool(g["meta_bn_jitter"]))
    model.load_state_dict(weights)
    return g, model.to(DEV), weights


@pytest.mark.parametrize("case", ["r2plus1d_small", "r3d_small", "r2plus1d_jitter", "r2plus1d_A"])
def test_train_mode_forward_backward(case):
    g, model, _ = build(case)
    x, z = case_inputs(g)
    model.train()
    bn_in = {}
    hooks = []
    for name, m in model.named_modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            hooks.append(m.register_forward_hook(
                lambda mod, inp, out, name=name: bn_in.__setitem__(name, inp[0].detach())))
    y = train.embed(model, x.to(DEV))
    loss = F.mse_loss(y, z.to(DEV))
    loss.backward()
    for h in hooks:
        h.remove()
    torch.cuda.synchronize()

    ynp = y.detach().cpu().numpy()
    assert rel_err(ynp, g["emb_f32"]) < EMB_TOL      # the stated bar
    assert rel_err(ynp, g["emb_f32"]) < TIGHT
    assert rel_err(ynp, g["emb_f64"]) < TIGHT
    assert abs(loss.item() - float(g["loss_f32"])) / float(g["loss_f32"]) < TIGHT

    # BatchNorm batch statistics of every BN input (the golden file lists them in the reference's
    # call order; the fused block runs the shortcut BN before conv2's BN, so compare by name)
    names = [str(n) for n in g["bn_names"]]
    assert sorted(bn_in.keys()) == sorted(names)
    mean = torch.cat([bn_in[n].double().mean(dim=(0, 2, 3, 4)) for n in names]).cpu().numpy()
    var = torch.cat([bn_in[n].double().var(dim=(0, 2, 3, 4), unbiased=False) for n in names]).cpu().numpy()
    assert np.abs(mean - g["bn_mean"]).max() < 1e-4 * (np.abs(g["bn_mean"]).max() + np.sqrt(g["bn_var"].max()))
    assert rel_err(var, g["bn_var"]) < 1e-4

    # which parameters receive gradients (dead Transformer encoder etc. must stay None)
    live = [k for k, p in model.named_parameters() if p.grad is not None]
    dead = [k for k, p in model.named_parameters() if p.grad is None]
    assert live == [str(k) for k in g["live_params"]]
    assert dead == [str(k) for k in g["dead_params"]]

    # per-parameter gradient norms + samples vs the fp64 oracle
    params = dict(model.named_parameters())
    worst = 0.0
    for name, norm, sample in zip(g["grad_names"], g["grad_norm_f64"], g["grad_sample_f64"]):
        gr = params[str(name)].grad.double().flatten()
        worst = max(worst, abs(gr.norm().item() - norm) / (norm + 1e-30))
        idx = sample_idx(gr.numel(), 16)
        got = gr[torch.from_numpy(idx).to(gr.device)].cpu().numpy()
        # individual tiny-gradient elements carry the fp32 BN-backward cancellation noise (SURVEY
        # section 4: fp32 CPU vs fp64 CPU is already ~1e-2 rel-L2), so compare the sample vector in L2
        assert rel_l2(got, sample[:len(idx)]) < 0.1, str(name)
    assert worst < 3e-2, f"worst grad-norm deviation {worst:.3e}"


@pytest.mark.parametrize("case", ["r2plus1d_small", "r2plus1d_jitter"])
def test_stage_statistics(case):
    g, model, _ = build(case)
    x, _ = case_inputs(g)
    model.train()
    trunk = model.model
    with torch.no_grad():
        t = x.reshape(x.shape[0], *x.shape[2:]).to(DEV)
        feats = [trunk.stem(t)]
        for i in range(1, 5):
            feats.append(getattr(trunk, f"layer{i}")(feats[-1]))
    for name, f in zip(["stem", "layer1", "layer2", "layer3", "layer4"], feats):
        flat = f.flatten()
        assert abs(flat.double().mean().item() - float(g[f"stage_{name}_mean"])) < 1e-4 * float(g[f"stage_{name}_absmean"])
        assert abs(flat.double().abs().mean().item() / float(g[f"stage_{name}_absmean"]) - 1) < 1e-4
        idx = torch.from_numpy(sample_idx(flat.numel())).to(DEV)
        assert rel_err(flat[idx].cpu().numpy(), g[f"stage_{name}_sample"]) < 5e-4, name


@pytest.mark.parametrize("case", ["r2plus1d_small", "r2plus1d_jitter"])
def test_adam_steps_running_stats_and_eval(case):
    g, model, weights = build(case)
    x, z = case_inputs(g)
    xd, zd = x.to(DEV), z.to(DEV)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = torch.nn.MSELoss()
    _, l1 = train.train_step(model, opt, crit, xd, zd)
    sd = model.state_dict()
    rm = torch.cat([sd[k].flatten() for k in sd if k.endswith("running_mean")]).cpu().numpy()
    rv = torch.cat([sd[k].flatten() for k in sd if k.endswith("running_var")]).cpu().numpy()
    assert abs(l1.item() / float(g["step1_loss"]) - 1) < TIGHT
    assert np.abs(rm - g["running_mean_after1"]).max() < 1e-4 * (np.abs(g["running_mean_after1"]).max() + 1)
    assert rel_err(rv, g["running_var_after1"]) < 1e-4
    nbt = [int(sd[k]) for k in sd if k.endswith("num_batches_tracked")]
    assert set(nbt) == {1}
    # second step: Adam's first update is ~lr*sign(g), so sign flips of tiny grads move the loss;
    # sanity only
    _, l2 = train.train_step(model, opt, crit, xd, zd)
    assert abs(l2.item() / float(g["step2_loss"]) - 1) < 0.2

    model.load_state_dict(weights)
    model.eval()
    with torch.no_grad():
        ye = train.embed(model, xd)
    assert rel_err(ye.cpu().numpy(), g["emb_eval_f32"]) < TIGHT
    if "emb_eval_t32_f32" in g:          # config E geometry: 32-frame clip through eval-mode BN
        x32 = synthetic.synthetic_clips(1, 32, int(g["meta_size"]), seed=99)
        with torch.no_grad():
            y32 = train.embed(model, x32.to(DEV))
        assert rel_err(y32.cpu().numpy(), g["emb_eval_t32_f32"]) < TIGHT


def test_c3d_eval_forward_backward():
    g, model, _ = build("c3d_eval")
    x, z = case_inputs(g)
    model.eval()                                   # dropout off
    y = train.embed(model, x.to(DEV))
    loss = F.mse_loss(y, z.to(DEV))
    loss.backward()
    assert rel_err(y.detach().cpu().numpy(), g["emb_f32"]) < TIGHT
    assert abs(loss.item() / float(g["loss_f32"]) - 1) < TIGHT
    params = dict(model.named_parameters())
    for name, norm in zip(g["grad_names"], g["grad_norm_f64"]):
        gr = params[str(name)].grad
        assert gr is not None, name
        assert abs(gr.double().norm().item() - norm) <= 5e-3 * norm + 1e-12, str(name)      # fp32 vs fp64 oracle
    assert [k for k, p in model.named_parameters() if p.grad is None] == [str(k) for k in g["dead_params"]]


def test_fixconvs_freezes_trunk_and_skips_its_gradients():
    model = network.get_network(make_opt("r2plus1d_18", fixconvs=True)).to(DEV)
    x = synthetic.synthetic_clips(1, 4, 32).to(DEV)
    y, _ = model(x)
    y.sum().backward()
    assert all(p.grad is None for p in model.model.parameters())
    assert all(p.grad is not None for p in model.output2emb_proj.parameters())


def test_evaluate_protocol_on_gpu():
    g, model, _ = build("r2plus1d_small")
    classes = synthetic.class_table(51, seed=77)
    batches = []
    for i in range(3):
        x = synthetic.synthetic_clips(2, 8, 56, seed=500 + i)
        labels, z = synthetic.synthetic_targets(2, 51, seed=77, rank=i)
        if i == 1:
            labels[0] = -1                                     # broken sample is dropped (main.py:246)
        batches.append((x, labels, z, torch.arange(2)))
    res = train.evaluate(model, batches, classes)
    assert res["n"] == 5
    assert 0.0 <= res["accuracy"] <= res["accuracy_top5"] <= 100.0


def test_gradient_sync_on_rccl_single_rank():
    """The RCCL code path of ddp.GradientSync (side stream, events, bucket pack/unpack) on the one
    GPU this box has: world_size 1, so averaged gradients must equal the local ones."""
    import os
    import torch.distributed as dist
    from zeroshotvideoclassification_amd import ddp
    if dist.is_initialized():
        pytest.skip("process group already initialised")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = "29531"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        g, model, _ = build("r2plus1d_small")
        x, z = case_inputs(g)
        xd, zd = x.to(DEV), z.to(DEV)
        model.train()
        crit = torch.nn.MSELoss()
        # reference gradients without the sync
        y = train.embed(model, xd)
        crit(y, zd).backward()
        ref = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        rs = {k: v.clone() for k, v in model.state_dict().items() if "running" in k}
        model.zero_grad(set_to_none=True)
        model.load_state_dict({**model.state_dict(), **rs})

        sync = ddp.GradientSync(model, bucket_bytes=8 * 1024 * 1024)
        opt = torch.optim.Adam(model.parameters(), lr=0.0)              # lr 0: parameters stay put
        for step in range(2):                                            # discovery step, then overlapped step
            train.train_step(model, opt, crit, xd, zd, sync)
            torch.cuda.synchronize()
            got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
            assert got.keys() == ref.keys()
            for k in ref:
                assert torch.allclose(got[k], ref[k], rtol=1e-5, atol=1e-9), (step, k)
        assert sync.live_parameter_count == len(ref) == len(g["live_params"])   # dead tensors excluded
        assert sum(sync.bucket_sizes) == 31716681
        assert len(sync.bucket_sizes) >= 4
        first = sync._buckets[0].params[0]
        assert any(first is p for p in model.output2emb_proj.parameters())   # head first, stem last
        sync.remove()
    finally:
        dist.destroy_process_group()


def test_full_size_batch22_forward_matches_oracle_and_is_deterministic():
    """BASELINE configs[1] geometry (22 clips of 3x16x112x112, train-mode BN): forward + loss against
    the CPU oracle run here on the same weights and clips, run-to-run bitwise determinism of forward
    and gradients, and the size-independent property that eval-mode embeddings of a clip do not
    depend on its batch mates."""
    from oracle import restatement as R
    opt = make_opt("r2plus1d_18")
    model = network.get_network(opt)
    weights = synthetic.keyed_state_dict(model.state_dict(), seed=0, bn_jitter=True)
    model.load_state_dict(weights)
    oracle = R.oracle_network(opt)
    oracle.load_state_dict(weights)
    x = synthetic.synthetic_clips(22, 16, 112)
    _, z = synthetic.synthetic_targets(22)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    oracle.train()
    with torch.no_grad():
        y_ref = R.embed(oracle, x)
        loss_ref = F.mse_loss(y_ref, z)

    model.to(DEV).train()
    xd, zd = x.to(DEV), z.to(DEV)
    runs = []
    for _ in range(2):
        model.load_state_dict(weights)                     # reset running statistics
        model.zero_grad(set_to_none=True)
        y = train.embed(model, xd)
        loss = F.mse_loss(y, zd)
        loss.backward()
        runs.append((y.detach().clone(), loss.detach().clone(),
                     model.model.stem[0].weight.grad.clone(), model.model.layer1[0].conv1[0][0].weight.grad.clone()))
    assert rel_err(runs[0][0].cpu().numpy(), y_ref.numpy()) < TIGHT
    assert abs(runs[0][1].item() / loss_ref.item() - 1) < TIGHT
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b), "two identical steps must agree bit for bit"

    model.eval()
    with torch.no_grad():
        full = train.embed(model, xd)
        for i in (0, 7, 21):
            single = train.embed(model, xd[i:i + 1])
            assert rel_err(single.cpu().numpy(), full[i:i + 1].cpu().numpy()) < 1e-5


def _oracle_gradients(opt, weights, x, z, dtype, train_mode):
    from oracle import restatement as R
    oracle = R.oracle_network(opt).to(dtype)
    oracle.load_state_dict({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in weights.items()})
    oracle.train(train_mode)
    y = R.embed(oracle, x.to(dtype))
    loss = F.mse_loss(y, z.to(dtype))
    loss.backward()
    grads = {k: p.grad.detach().double().numpy() for k, p in oracle.named_parameters() if p.grad is not None}
    return y.detach().double().numpy(), float(loss.item()), grads


@pytest.mark.parametrize("net,n,train_mode", [("r2plus1d_18", 22, True), ("c3d", 4, False)])
def test_full_size_gradients_match_the_oracle_per_parameter(net, n, train_mode):
    """BASELINE configs[1] / [3] at FULL size (22 clips 3x16x112x112 train-mode R(2+1)D-18; 4 clips C3D,
    dropout off): EVERY live parameter's gradient against the CPU oracle run here on the same weights and
    clips -- the split-K, slice-count, tile and XCD-order choices of the kernels differ between the small
    golden cases and this size.  Bars: embeddings / loss 1e-4; per-parameter rel-L2 5e-2 vs the fp32 oracle
    and 3e-2 vs the fp64 oracle (fp32 CPU itself sits ~1e-2 from fp64: BN-backward cancellation)."""
    opt = make_opt(net)
    model = network.get_network(opt)
    weights = synthetic.keyed_state_dict(model.state_dict(), seed=0, bn_jitter=True)
    model.load_state_dict(weights)
    x = synthetic.synthetic_clips(n, 16, 112)
    _, z = synthetic.synthetic_targets(n)
    torch.set_num_threads(min(16, torch.get_num_threads()))

    model.to(DEV).train(train_mode)
    y = train.embed(model, x.to(DEV))
    loss = F.mse_loss(y, z.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    got = {k: p.grad.detach().cpu().double().numpy() for k, p in model.named_parameters() if p.grad is not None}
    y_np = y.detach().cpu().numpy()

    for dtype, tol in ((torch.float32, 5e-2), (torch.float64, 3e-2)):
        y_ref, loss_ref, ref = _oracle_gradients(opt, weights, x, z, dtype, train_mode)
        assert rel_err(y_np, y_ref) < TIGHT
        assert abs(loss.item() / loss_ref - 1) < TIGHT
        assert sorted(got) == sorted(ref)                                   # same live / dead split
        worst = max((rel_l2(got[k], ref[k]), k) for k in ref)
        assert worst[0] < tol, (str(dtype), worst)
    assert len(got) == (115 if net == "r2plus1d_18" else 20)          # 115 of 193 (R(2+1)D-18 + head), 20 of 24 (fc7 / fc8 unused)


def test_c3d_at_the_benchmark_batch_matches_the_oracle_forward():
    """BASELINE configs[3] at the batch the bench runs (22 clips; the per-parameter gradient check above stays at 4): conv1's
    1.13 GB output, the tile / slice choices of every layer at N = 22.  Embeddings and loss against the CPU oracle on the same
    weights and clips (eval mode: dropout off), and the backward at that size must produce finite gradients for exactly the live set."""
    opt = make_opt("c3d")
    model = network.get_network(opt)
    weights = synthetic.keyed_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(weights)
    n = 22
    x = synthetic.synthetic_clips(n, 16, 112)
    _, z = synthetic.synthetic_targets(n)
    from oracle import restatement as R
    torch.set_num_threads(min(16, torch.get_num_threads()))
    oracle = R.oracle_network(opt)
    oracle.load_state_dict(weights)
    oracle.eval()
    with torch.no_grad():
        y_ref = R.embed(oracle, x)
        loss_ref = F.mse_loss(y_ref, z)
    model.to(DEV).eval()
    y = train.embed(model, x.to(DEV))
    loss = F.mse_loss(y, z.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert y.shape == (n, 300)
    assert rel_err(y.detach().cpu().numpy(), y_ref.numpy()) < TIGHT
    assert abs(loss.item() / loss_ref.item() - 1) < TIGHT
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert len(grads) == 20 and all(torch.isfinite(g).all().item() for g in grads.values())


def test_mc3_18_trunk_matches_oracle_trunk():
    """resnet.mc3_18 (resnet.py:318-338) is not reachable through get_network but is part of the
    module surface: 3x3x3 first stage, 1x3x3 (Conv3DNoTemporal, shortcut stride (1,s,s)) afterwards."""
    from oracle import restatement as R
    from zeroshotvideoclassification_amd import resnet
    trunk = resnet.mc3_18()
    ref = R.video_trunk("mc3_18")()
    assert list(trunk.state_dict().keys()) == list(ref.state_dict().keys())
    weights = synthetic.keyed_state_dict(ref.state_dict(), seed=2, bn_jitter=True)
    trunk.load_state_dict(weights)
    ref.load_state_dict(weights)
    x = synthetic.synthetic_clips(2, 8, 48, seed=5).reshape(2, 3, 8, 48, 48)
    ref.train()
    pooled_ref, f_ref = ref(x)
    pooled_ref.sum().backward()
    trunk.to(DEV).train()
    pooled, f = trunk(x.to(DEV))
    pooled.sum().backward()
    assert f.shape == f_ref.shape == (2, 512, 8, 3, 3)                # no temporal striding after stage 1
    assert rel_err(pooled.detach().cpu().numpy(), pooled_ref.detach().numpy()) < TIGHT
    assert rel_err(f.detach().cpu().numpy(), f_ref.detach().numpy()) < 5e-4
    gr = dict(ref.named_parameters())
    for k, p in trunk.named_parameters():
        if gr[k].grad is None:
            assert p.grad is None, k
        else:
            assert rel_l2(p.grad.cpu().numpy(), gr[k].grad.numpy()) < 5e-2, k


def test_training_loop_converges_like_the_oracle():
    """30 Adam steps on a fixed synthetic batch (overfit): the loss must fall monotonically-ish, stay
    finite, and track the CPU oracle's trajectory (same weights, same data, fp32) within 10 % per step
    for the first steps and end below a fifth of where it started (sanity of the whole training path)."""
    from oracle import restatement as R
    opt = make_opt("r2plus1d_18")
    model = network.get_network(opt)
    weights = synthetic.keyed_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(weights)
    oracle = R.oracle_network(opt)
    oracle.load_state_dict(weights)
    x = synthetic.synthetic_clips(4, 4, 32)
    _, z = synthetic.synthetic_targets(4)
    model.to(DEV).train()
    oracle.train()
    opt_g = torch.optim.Adam(model.parameters(), lr=1e-3)
    opt_c = torch.optim.Adam(oracle.parameters(), lr=1e-3)
    crit = torch.nn.MSELoss()
    xd, zd = x.to(DEV), z.to(DEV)
    lg, lc = [], []
    for step in range(30):
        _, l = train.train_step(model, opt_g, crit, xd, zd)
        lg.append(l.item())
        if step < 6:
            _, l2 = R.train_step(oracle, opt_c, x, z)
            lc.append(l2.item())
    assert all(np.isfinite(lg))
    assert abs(lg[0] / lc[0] - 1) < 1e-4
    for a, b in zip(lg[:6], lc):
        assert abs(a / b - 1) < 0.10, (lg[:6], lc)
    assert lg[-1] < 0.2 * lg[0], lg


def test_identity_shortcut_gradient_is_added_in_the_dgrad_epilogue(monkeypatch):
    """BasicBlock with an identity shortcut: the linked path (ops.SkipLink: shortcut gradient added inside
    the first convolution's dgrad) must give the bitwise-same input gradient as the unlinked path
    (separate add), and the linked path must actually run."""
    from zeroshotvideoclassification_amd import ops, resnet
    torch.manual_seed(3)
    block = resnet.BasicBlock(64, 64, resnet.Conv2Plus1D).to(DEV).train()
    shape = (4, 64, 8, 76, 76)                 # enough voxels that the dgrad runs without split-K (else: no linking)
    x0 = torch.randn(shape, device=DEV)
    dy = torch.randn(shape, device=DEV)

    def run():
        x = x0.clone().requires_grad_(True)
        links = []
        orig = ops.SkipLink.__init__

        def spy(self):
            orig(self)
            links.append(self)
        monkeypatch.setattr(ops.SkipLink, "__init__", spy)
        y = block(x)
        y.backward(dy)
        monkeypatch.setattr(ops.SkipLink, "__init__", orig)
        return y.detach(), x.grad, [p.grad.clone() for p in block.parameters()], links

    y1, g1, p1, links = run()
    assert len(links) == 1 and links[0].armed and links[0].dres is None      # armed, and consumed by the dgrad
    block.zero_grad()
    monkeypatch.setenv("ZSV_NO_SKIP_FUSION", "1")
    y2, g2, p2, links2 = run()
    assert links2 == []
    assert torch.equal(y1, y2) and torch.equal(g1, g2)
    for a, b in zip(p1, p2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("inplanes,planes,shape", [(64, 128, (3, 64, 8, 56, 56)), (256, 512, (3, 256, 4, 14, 14)), (64, 128, (2, 64, 6, 20, 28))],
                         ids=["layer2_like", "layer4_like_k_parts", "t_not_a_multiple_of_four"])
def test_strided_shortcut_gradient_is_added_in_the_dgrad_epilogue(inplanes, planes, shape, monkeypatch):
    """BasicBlock with a strided 1x1x1 `downsample` (resnet.py:240-246): the linked path (ops.DownLink: the shortcut's input
    gradient in compact form, added inside the strided first convolution's dgrad) against the unlinked path (zero-filled
    full-size shortcut gradient + autograd's add): same forward, same parameter gradients bit for bit; the input gradient to
    rounding (the compact shortcut gradient is a stride-1 problem with its own tiling, and with K parts the add joins part 0)."""
    from zeroshotvideoclassification_amd import layers, ops, resnet
    torch.manual_seed(5)
    down = torch.nn.Sequential(layers.Conv3d(inplanes, planes, kernel_size=1, stride=(2, 2, 2), bias=False), layers.BatchNorm3d(planes))
    block = resnet.BasicBlock(inplanes, planes, resnet.Conv2Plus1D, stride=2, downsample=down).to(DEV).train()
    x0 = torch.randn(shape, device=DEV)

    def run():
        x = x0.clone().requires_grad_(True)
        links = []
        orig = ops.DownLink.__init__

        def spy(self, strides):
            orig(self, strides)
            links.append(self)
        monkeypatch.setattr(ops.DownLink, "__init__", spy)
        y = block(x)
        g = torch.Generator(device="cpu").manual_seed(9)
        y.backward(torch.randn(y.shape, generator=g).to(DEV))
        monkeypatch.setattr(ops.DownLink, "__init__", orig)
        return y.detach(), x.grad, [p.grad.clone() for p in block.parameters()], links

    y1, g1, p1, links = run()
    assert len(links) == 1 and links[0].armed and links[0].consumed and links[0].dsub is None      # parked, then collected
    block.zero_grad()
    monkeypatch.setenv("ZSV_NO_DOWN_FUSION", "1")
    y2, g2, p2, links2 = run()
    assert links2 == []
    assert torch.equal(y1, y2)
    for a, b in zip(p1, p2):
        assert torch.equal(a, b)
    err = (g1 - g2).abs().max().item()
    assert err <= 2e-6 * g2.abs().max().item(), (err, g2.abs().max().item())
    # the fused input gradient against torch CPU fp64 on the same block
    xr = x0.double().cpu().requires_grad_(True)
    blk = block.cpu().double()
    try:
        import torch.nn.functional as F
        c1s, bn_mid, _, c1t = list(blk.conv1[0])
        bn1 = blk.conv1[1]
        c2s, bn_mid2, _, c2t = list(blk.conv2[0])
        bn2 = blk.conv2[1]
        dconv, dbn = list(blk.downsample)

        def bn(m, t):
            return F.batch_norm(t, None, None, m.weight, m.bias, True, 0.0, m.eps)
        h = F.relu(bn(bn_mid, F.conv3d(xr, c1s.weight, None, c1s.stride, c1s.padding)))
        h = F.relu(bn(bn1, F.conv3d(h, c1t.weight, None, c1t.stride, c1t.padding)))
        h = F.relu(bn(bn_mid2, F.conv3d(h, c2s.weight, None, c2s.stride, c2s.padding)))
        h = bn(bn2, F.conv3d(h, c2t.weight, None, c2t.stride, c2t.padding))
        yr = F.relu(h + bn(dbn, F.conv3d(xr, dconv.weight, None, dconv.stride, dconv.padding)))
        g = torch.Generator(device="cpu").manual_seed(9)
        yr.backward(torch.randn(yr.shape, generator=g).double())
        # (relative L2: a ReLU input within rounding of zero flips its mask and moves single voxels by O(0.1))
        e64 = ((g1.double().cpu() - xr.grad).norm() / xr.grad.norm()).item()
        assert e64 <= 2e-3, e64
    finally:
        block.float().to(DEV)


@pytest.mark.parametrize("n,t,h,w", [(2, 8, 72, 88), (3, 10, 50, 50)])
def test_odd_clip_sizes_forward_backward_match_the_oracle(n, t, h, w):
    """Clip sizes whose feature maps are odd / not multiples of the kernels' vector widths (voxel counts
    not divisible by 4 or 16: the register-staged wgrad, scalar epilogues, ragged tiles): train-mode
    forward, loss and every parameter gradient against the CPU oracle (fp32) on the same weights."""
    from oracle import restatement as R
    opt = make_opt("r2plus1d_18")
    model = network.get_network(opt)
    weights = synthetic.keyed_state_dict(model.state_dict(), seed=4, bn_jitter=True)
    model.load_state_dict(weights)
    oracle = R.oracle_network(opt)
    oracle.load_state_dict(weights)
    g = torch.Generator().manual_seed(h * w + t)
    x = (torch.randint(0, 256, (n, 1, 3, t, h, w), generator=g).float() / 255.0 - 1.0) / 2.0
    _, z = synthetic.synthetic_targets(n)
    oracle.train()
    y_ref = R.embed(oracle, x)
    loss_ref = F.mse_loss(y_ref, z)
    loss_ref.backward()
    model.to(DEV).train()
    y = train.embed(model, x.to(DEV))
    loss = F.mse_loss(y, z.to(DEV))
    loss.backward()
    assert rel_err(y.detach().cpu().numpy(), y_ref.detach().numpy()) < TIGHT
    assert abs(loss.item() / loss_ref.item() - 1) < TIGHT
    ref_grads = {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(got) == set(ref_grads)
    worst = 0.0
    for k, gr in ref_grads.items():
        a, b = got[k].detach().cpu().double().flatten(), gr.double().flatten()
        worst = max(worst, ((a - b).norm() / (b.norm() + 1e-30)).item())
    # fp32 GPU vs fp32 CPU, different summation orders through ~40 layers of BatchNorm backward
    assert worst < 5e-2, f"worst per-parameter gradient rel-L2 {worst:.3e}"


def test_weight_gradients_on_the_side_stream_are_the_same_bits(monkeypatch):
    """ops: every convolution's wgrad runs on a side HIP stream (joined by an autograd-engine callback at the end of
    backward).  Same kernels, same order per tensor: `.grad` read right after `loss.backward()` must equal, bit for bit,
    the gradients of the single-stream run -- also when the optimizer step follows immediately."""
    g, model, weights = build("r2plus1d_small")
    x, z = case_inputs(g)
    xd, zd = x.to(DEV), z.to(DEV)

    def grads():
        model.load_state_dict(weights)
        model.train()
        model.zero_grad(set_to_none=True)
        F.mse_loss(train.embed(model, xd), zd).backward()
        return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}     # (clone: main stream, after the join)

    monkeypatch.setenv("ZSV_WGRAD_STREAM", "0")
    ref = grads()
    monkeypatch.setenv("ZSV_WGRAD_STREAM", "1")
    for _ in range(2):
        got = grads()
        assert sorted(got) == sorted(ref)
        for k in ref:
            assert torch.equal(got[k], ref[k]), k
    # and through a full step: parameters after Adam agree too
    def step():
        model.load_state_dict(weights)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        train.train_step(model, opt, torch.nn.MSELoss(), xd, zd)
        return {k: v.clone() for k, v in model.state_dict().items()}

    a = step()
    monkeypatch.setenv("ZSV_WGRAD_STREAM", "0")
    b = step()
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_folded_batchnorm_is_the_same_bits_at_model_level(monkeypatch):
    """The mid BatchNorm + ReLU of every stride-1 Conv2Plus1D is folded into its temporal convolution (resnet._run_chain);
    ZSV_NO_BN_FUSION=1 runs the separate normalise pass instead.  One training step either way: loss, embeddings, every
    gradient, running statistics and the updated parameters agree bit for bit."""
    g, model, weights = build("r2plus1d_A")
    x, z = case_inputs(g)
    xd, zd = x.to(DEV), z.to(DEV)

    def step():
        model.load_state_dict(weights)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        y, loss = train.train_step(model, opt, torch.nn.MSELoss(), xd, zd)
        grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        return y.clone(), loss.clone(), grads, {k: v.clone() for k, v in model.state_dict().items()}

    from zeroshotvideoclassification_amd import ops
    assert ops.conv_pre_supported((2, 144, 16, 56, 56), (64, 144, 3, 1, 1), 1, (1, 0, 0))       # layer1's pairs take the folded path
    ya, la, ga, sa = step()
    monkeypatch.setenv("ZSV_NO_BN_FUSION", "1")
    yb, lb, gb, sb = step()
    assert torch.equal(ya, yb) and torch.equal(la, lb)
    assert sorted(ga) == sorted(gb)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_gradient_accumulation_with_the_wgrad_side_stream():
    """`zero_grad(set_to_none=False)` / two backward passes into the same `.grad`: autograd then adds each weight gradient
    in place on the backward stream, which must wait for the side stream that computed it."""
    g, model, weights = build("r2plus1d_small")
    x, z = case_inputs(g)
    xd, zd = x.to(DEV), z.to(DEV)
    model.load_state_dict(weights)
    model.train()
    model.zero_grad(set_to_none=True)
    F.mse_loss(train.embed(model, xd), zd).backward()
    once = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.load_state_dict(weights)
    F.mse_loss(train.embed(model, xd), zd).backward()            # accumulates into the existing .grad
    torch.cuda.synchronize()
    for k, p in model.named_parameters():
        if p.grad is not None and "running" not in k:
            ref = once[k] * 2
            err = (p.grad - ref).abs().max().item()
            assert err <= 1e-5 * (ref.abs().max().item() + 1e-12), (k, err)


def test_shared_weights_in_one_graph_with_the_wgrad_side_stream(monkeypatch):
    """One weight, two gradients inside ONE backward pass (the model applied to two clip batches before backward: siamese /
    multi-clip forwards, tied weights): autograd sums dw1 + dw2 on the backward stream as soon as the second arrives, so that
    stream has to wait for the side stream still writing them.  Also with a tensor hook on a weight (runs on dw right away).
    Must equal the single-stream run bit for bit."""
    g, model, weights = build("r2plus1d_small")
    x, z = case_inputs(g)
    xd, zd = x.to(DEV), z.to(DEV)
    xd2 = (xd * 0.5 - 0.1).contiguous()
    hooked = model.model.layer1[0].conv1[0][0].weight
    seen = []

    def grads():
        model.load_state_dict(weights)
        model.train()
        model.zero_grad(set_to_none=True)
        seen.clear()
        h = hooked.register_hook(lambda gw: seen.append(gw.abs().sum().clone()))       # reads dw on the backward stream
        try:
            loss = F.mse_loss(train.embed(model, xd), zd) + 0.5 * F.mse_loss(train.embed(model, xd2), zd)
            loss.backward()
        finally:
            h.remove()
        return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, seen[0].item()

    monkeypatch.setenv("ZSV_WGRAD_STREAM", "0")
    ref, ref_hook = grads()
    monkeypatch.setenv("ZSV_WGRAD_STREAM", "1")
    for _ in range(3):
        got, got_hook = grads()
        assert sorted(got) == sorted(ref)
        for k in ref:
            assert torch.equal(got[k], ref[k]), k
        assert got_hook == ref_hook


def test_forward_hooks_see_the_reference_tensors():
    """A forward hook on the mid BatchNorm3d / ReLU of a Conv2Plus1D (resnet.py:46-52) must receive the normalised / activated
    tensor as with the reference's modules: hooked triples run the separate passes instead of the fold (which never materialises
    it).  The model output does not change (the fold is bit-identical)."""
    g, model, weights = build("r2plus1d_small")
    x, _ = case_inputs(g)
    xd = x.to(DEV)
    model.train()
    model.load_state_dict(weights)
    with torch.no_grad():
        ref = train.embed(model, xd).clone()
    pair = model.model.layer1[0].conv1[0]                    # Conv2Plus1D: [spatial conv, BatchNorm3d, ReLU, temporal conv]
    seen = {}
    h1 = pair[1].register_forward_hook(lambda m, i, o: seen.__setitem__("bn", o))
    h2 = pair[2].register_forward_hook(lambda m, i, o: seen.__setitem__("relu", o))
    try:
        model.load_state_dict(weights)
        with torch.no_grad():
            out = train.embed(model, xd)
    finally:
        h1.remove()
        h2.remove()
    assert torch.equal(out, ref)
    bn, act = seen["bn"], seen["relu"]
    assert torch.is_tensor(bn) and torch.is_tensor(act) and bn.shape == act.shape and bn.dim() == 5 and bn.shape[1] == pair[1].num_features
    assert torch.equal(act, torch.relu(bn))
    m = bn.mean(dim=(0, 2, 3, 4))
    assert (m - pair[1].bias.detach()).abs().max().item() < 1e-4     # normalised: per-channel mean = beta


def test_weight_panel_cache_is_the_same_bits_and_follows_the_weights(monkeypatch):
    """ops._PanelCache: every (weight, geometry, direction) keeps its packed panel; when the optimizer has changed the weights the
    first convolution of the next step re-packs ALL panels in one zsv_pack_multi launch and the calls skip their pack launches.
    Three Adam steps with the cache equal three steps without it (ZSV_NO_PANEL_CACHE=1) bit for bit -- loss, every gradient, every
    updated parameter -- and a weight written behind autograd's back is picked up after _lib.note_raw_write()."""
    from zeroshotvideoclassification_amd import _lib, ops
    g, model, weights = build("r2plus1d_small")
    x, z = case_inputs(g)
    xd, zd = x.to(DEV), z.to(DEV)

    def three_steps():
        model.load_state_dict(weights)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        out = []
        for _ in range(3):
            y, loss = train.train_step(model, opt, torch.nn.MSELoss(), xd, zd)
            out.append((y.clone(), loss.clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
        return out, {k: v.clone() for k, v in model.state_dict().items()}

    ops.invalidate_panels()
    a, sa = three_steps()
    cache = ops._PANEL_CACHES[torch.device(DEV).index or 0]
    live = [e for e in cache.entries.values() if e is not None]
    assert len(live) >= 60                                    # forward + input-gradient panels of the 37 convolutions (stems have none)
    repacks_before = cache.repacks
    three_steps()
    # steady state: load_state_dict + 3 optimizer steps = at most 4 weight versions -> a handful of multi-pack launches, not 74 per step
    assert cache.repacks - repacks_before <= 8
    monkeypatch.setenv("ZSV_NO_PANEL_CACHE", "1")
    b, sb = three_steps()
    monkeypatch.delenv("ZSV_NO_PANEL_CACHE")
    for (ya, la, ga), (yb, lb, gb) in zip(a, b):
        assert torch.equal(ya, yb) and torch.equal(la, lb)
        assert sorted(ga) == sorted(gb)
        for k in ga:
            assert torch.equal(ga[k], gb[k]), k
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    # a raw write (no version bump) followed by note_raw_write(): the next forward uses the new weights (with the cache switched on
    # for inference; by default calls without autograd pack per call, so that `.data` edits of an eval model need no announcement)
    model.load_state_dict(weights)
    model.eval()
    w = model.model.layer1[0].conv1[0][0].weight
    with torch.no_grad():
        y0 = train.embed(model, xd).clone()
        w0 = w.data.clone()
        w.data.mul_(1.5)
        w15 = w.data.clone()
        y1_plain = train.embed(model, xd).clone()             # (no note_raw_write, no cache: already right)
        monkeypatch.setenv("ZSV_PANEL_CACHE_EVAL", "1")
        train.embed(model, xd)                                # panels of the 1.5x weights are cached now
        w.data.copy_(w0 * 2.0)
        _lib.note_raw_write()
        y2 = train.embed(model, xd).clone()
        monkeypatch.delenv("ZSV_PANEL_CACHE_EVAL")
        monkeypatch.setenv("ZSV_NO_PANEL_CACHE", "1")
        y2_ref = train.embed(model, xd).clone()
        w.data.copy_(w15)
        y1_ref = train.embed(model, xd).clone()
    assert not torch.equal(y0, y1_plain) and torch.equal(y1_plain, y1_ref)
    assert not torch.equal(y1_plain, y2) and torch.equal(y2, y2_ref)


def test_weight_panel_cache_is_bounded(monkeypatch):
    """A caller that keeps changing clip shapes would add a panel per (weight, geometry) for ever: past ZSV_PANEL_CACHE_MB the cache is
    emptied and refills with the shapes in use; results do not change."""
    from zeroshotvideoclassification_amd import ops
    g, model, weights = build("r2plus1d_small")
    model.load_state_dict(weights)
    model.eval()
    ops.invalidate_panels()
    monkeypatch.setenv("ZSV_PANEL_CACHE_MB", "1")
    monkeypatch.setenv("ZSV_PANEL_CACHE_EVAL", "1")           # (the cache is off without autograd by default)
    outs = {}
    with torch.no_grad():
        for rep in range(2):
            for n in (1, 2, 3):
                x = torch.randn(n, 1, 3, 8, 32, 32, generator=torch.Generator().manual_seed(n)).to(DEV)
                y = train.embed(model, x).clone()
                if rep:
                    assert torch.equal(y, outs[n]), n
                outs[n] = y
    cache = ops._PANEL_CACHES[torch.device(DEV).index or 0]
    assert cache.limit == 1 << 20
    assert cache.nbytes <= cache.limit + max(e.nbytes for e in cache.entries.values() if e is not None) * 80      # (one network's worth past the limit at most)
    ops.invalidate_panels()


def test_panel_verify_mode_catches_an_unannounced_raw_write(monkeypatch):
    """ZSV_PANEL_VERIFY=1 (debug): every cached weight panel is re-packed into a scratch buffer and compared before use, so a weight
    edited through a `.data` alias without `_lib.note_raw_write()` raises instead of silently running forward / dgrad on the stale
    panel.  Unmodified weights pass, and an announced write passes too."""
    from zeroshotvideoclassification_amd import _lib, ops
    g, model, weights = build("r2plus1d_small")
    x, z = case_inputs(g)
    xd, zd = x.to(DEV), z.to(DEV)
    model.load_state_dict(weights)
    model.train()
    ops.invalidate_panels()
    monkeypatch.setenv("ZSV_PANEL_VERIFY", "1")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    train.train_step(model, opt, torch.nn.MSELoss(), xd, zd)          # panels are built and verified against their weights
    train.train_step(model, opt, torch.nn.MSELoss(), xd, zd)          # ... and again after the optimizer moved them
    # (the first forward after an optimizer step re-packs every panel from whatever the weights hold: bring the panels up to date
    # first, so that the edit below really is one the cache cannot see)
    model.zero_grad(set_to_none=True)
    torch.nn.MSELoss()(train.embed(model, xd), zd).backward()
    ops.join_wgrad_streams()
    w = model.model.layer2[1].conv1[0][0].weight
    w.data.mul_(1.25)                                                 # behind autograd's back: no version bump, no note
    model.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match="ZSV_PANEL_VERIFY"):
        criterion = torch.nn.MSELoss()
        criterion(train.embed(model, xd), zd).backward()
    _lib.note_raw_write()                                             # announced: the panels are re-packed, the check passes
    model.zero_grad(set_to_none=True)
    torch.nn.MSELoss()(train.embed(model, xd), zd).backward()
    ops.join_wgrad_streams()
    torch.cuda.synchronize()
    ops.invalidate_panels()


def test_post_accumulate_grad_hook_sees_the_finished_weight_gradient():
    """a user's `register_post_accumulate_grad_hook` (optimizer-in-backward, clipping, logging) reads `p.grad`
    on the backward stream while the weight-gradient kernel may still run on the side stream.  `ops._dw_read_early` now joins the
    side stream for such parameters: the copy the hook takes equals the gradient after the pass, for every convolution weight."""
    g, model, weights = build("r2plus1d_small")
    x, z = case_inputs(g)
    xd, zd = x.to(DEV), z.to(DEV)
    model.load_state_dict(weights)
    model.train()
    seen = {}
    names = {id(p): k for k, p in model.named_parameters()}
    hooks = [p.register_post_accumulate_grad_hook(lambda p: seen.__setitem__(names[id(p)], p.grad.clone()))
             for k, p in model.named_parameters() if p.dim() == 5]
    try:
        model.zero_grad(set_to_none=True)
        torch.nn.MSELoss()(train.embed(model, xd), zd).backward()
        from zeroshotvideoclassification_amd import ops
        ops.join_wgrad_streams()
        torch.cuda.synchronize()
        assert len(seen) >= 37
        for k, p in model.named_parameters():
            if k in seen:
                assert torch.equal(seen[k], p.grad), k
    finally:
        for h in hooks:
            h.remove()
    # the data-parallel hook joins the side stream itself and says so: it must not cost the overlap
    from zeroshotvideoclassification_amd import ddp, ops
    sync = ddp.GradientSync(model, local=True)
    try:
        w = model.model.layer1[0].conv1[0][0].weight
        w.grad = None
        assert not any(not getattr(h, "_zsv_joins_wgrad", False) for h in w._post_accumulate_grad_hooks.values())
    finally:
        sync.remove()


def test_frozen_trunk_forward_under_grad_mode_does_not_use_cached_panels(monkeypatch):
    """with grad mode on but nothing to record (frozen weights, input without requires_grad) a forward is
    inference for the panel cache: every call packs from the weights as they are, so a `.data` edit needs no announcement."""
    from zeroshotvideoclassification_amd import ops
    g, model, weights = build("r2plus1d_small")
    x, _ = case_inputs(g)
    xd = x.to(DEV)
    model.load_state_dict(weights)
    model.eval()
    for p in model.parameters():
        p.requires_grad_(False)
    ops.invalidate_panels()
    assert torch.is_grad_enabled()
    y0 = train.embed(model, xd).clone()
    cache = ops._PANEL_CACHES.get(torch.device(DEV).index or 0)
    assert cache is None or not [e for e in cache.entries.values() if e is not None]
    w = model.model.layer1[0].conv1[0][0].weight
    w.data.mul_(1.5)
    y1 = train.embed(model, xd).clone()
    with torch.no_grad():
        y1_ref = train.embed(model, xd).clone()
    assert not torch.equal(y0, y1) and torch.equal(y1, y1_ref)


def test_step_pacer_bounds_the_host_lead_and_changes_no_result():
    """train.StepPacer(depth): the host never has more than `depth` unfinished steps behind the one it is queueing; the losses of
    paced and unpaced runs are the same bits."""
    g, model, weights = build("r2plus1d_small")
    x, z = case_inputs(g)
    xd, zd = x.to(DEV), z.to(DEV)

    def run(pacer):
        model.load_state_dict(weights)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        losses = []
        for _ in range(6):
            _, loss = train.train_step(model, opt, torch.nn.MSELoss(), xd, zd, pacer=pacer)
            if pacer is not None:
                assert len(pacer.marks) <= pacer.depth
                unfinished = sum(0 if ev.query() else 1 for ev in pacer.marks)
                assert unfinished <= pacer.depth
            losses.append(loss)
        torch.cuda.synchronize()
        return torch.stack(losses)

    a = run(None)
    b = run(train.StepPacer(1))
    c = run(train.StepPacer(2))
    assert torch.equal(a, b) and torch.equal(a, c)
