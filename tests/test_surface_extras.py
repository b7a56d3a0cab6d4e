"""Module surface that ``get_network`` does not reach, against outputs of the reference's own classes
(``tests/golden/surface_extras.npz``, ``oracle/make_golden.py::run_surface_extras``): ``resnet.Bottleneck`` inside
``VideoResNet`` (resnet.py:116-162,190-281) and the original ``network.ResNet18`` head (network.py:50-80); plus C3D in
TRAIN mode (network.py:166-167) with the dropout mask shared between the HIP path and the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import load_golden, make_opt, rel_err, rel_l2
from zeroshotvideoclassification_amd import network, resnet, synthetic

DEV = "cuda"


def test_state_dict_keys_of_the_extra_surface_match_the_reference():
    g = load_golden("surface_extras")
    trunk = resnet.VideoResNet(block=resnet.Bottleneck, conv_makers=[resnet.Conv2Plus1D] * 4, layers=[1, 1, 1, 1],
                               stem=resnet.R2Plus1dStem)
    assert list(trunk.state_dict().keys()) == [str(k) for k in g["bottleneck_keys"]]
    head = network.ResNet18(resnet.r2plus1d_18, fixconvs=False, nopretrained=False)
    assert list(head.state_dict().keys()) == [str(k) for k in g["resnet18_keys"]]


@pytest.mark.gpu
def test_bottleneck_trunk_matches_the_reference():
    g = load_golden("surface_extras")
    trunk = resnet.VideoResNet(block=resnet.Bottleneck, conv_makers=[resnet.Conv2Plus1D] * 4, layers=[1, 1, 1, 1],
                               stem=resnet.R2Plus1dStem)
    trunk.load_state_dict(synthetic.keyed_state_dict(trunk.state_dict(), seed=3, bn_jitter=True))
    trunk.to(DEV).train()
    x = synthetic.synthetic_clips(2, 4, 32, seed=21).reshape(2, 3, 4, 32, 32).to(DEV)
    pooled, f = trunk(x)
    assert tuple(f.shape) == tuple(int(v) for v in g["bottleneck_feature_shape"])
    assert rel_err(pooled.detach().cpu().numpy(), g["bottleneck_pooled_f32"]) < 1e-4
    assert rel_err(pooled.detach().cpu().numpy(), g["bottleneck_pooled_f64"]) < 1e-4
    (pooled * pooled).sum().backward()
    params = dict(trunk.named_parameters())
    live = [k for k, p in params.items() if p.grad is not None]
    assert sorted(live) == sorted(str(k) for k in g["bottleneck_grad_names"])
    for k, norm in zip(g["bottleneck_grad_names"], g["bottleneck_grad_norm_f64"]):
        got = params[str(k)].grad.double().norm().item()
        assert abs(got - norm) <= 3e-2 * norm + 1e-9, (str(k), got, norm)


@pytest.mark.gpu
def test_resnet18_head_matches_the_reference():
    g = load_golden("surface_extras")
    head = network.ResNet18(resnet.r2plus1d_18, fixconvs=False, nopretrained=False)
    head.load_state_dict(synthetic.keyed_state_dict(head.state_dict(), seed=4, bn_jitter=True))
    head.to(DEV).eval()
    x = synthetic.synthetic_clips(2, 4, 32, seed=21).to(DEV)
    y = head(x)
    assert rel_err(y.detach().cpu().numpy(), g["resnet18_emb_eval_f32"]) < 1e-4
    # gradients: train mode (batch statistics) with the RNG-dependent Dropout(0.05) replaced by the identity on both sides
    # (the HIP BatchNorm has no eval-mode backward: the reference evaluates under no_grad, main.py:230)
    head.dropout = torch.nn.Identity()
    head.train()
    y = head(x)
    assert rel_err(y.detach().cpu().numpy(), g["resnet18_emb_train_nodrop_f64"]) < 1e-4
    (y * synthetic.synthetic_targets(2)[1].to(DEV)).sum().backward()        # (sum(y^2) is constant for unit-norm rows)
    params = dict(head.named_parameters())
    assert sorted(k for k, p in params.items() if p.grad is not None) == sorted(str(k) for k in g["resnet18_grad_names"])
    for k, norm in zip(g["resnet18_grad_names"], g["resnet18_grad_norm_f64"]):
        got = params[str(k)].grad.double().norm().item()
        assert abs(got - norm) <= 3e-2 * norm + 1e-9, (str(k), got, norm)


class _SharedMaskDropout(torch.nn.Module):
    """nn.Dropout(p) with the Bernoulli mask drawn once on the CPU from a seed, so two implementations drop the
    same activations (the reference's train-mode C3D is otherwise RNG-dependent)."""

    def __init__(self, p, shape, seed):
        super().__init__()
        keep = (torch.rand(shape, generator=torch.Generator().manual_seed(seed)) >= p).float() / (1.0 - p)
        self.register_buffer("keep", keep)

    def forward(self, x):
        return x * self.keep.to(x.dtype) if self.training else x


@pytest.mark.gpu
def test_c3d_train_mode_with_a_shared_dropout_mask():
    """C3D forward + backward in TRAIN mode (network.py:147-179 with Dropout(0.10) after fc6) at full clip size:
    embeddings, loss and every live parameter's gradient against the CPU oracle."""
    from oracle import restatement as R
    opt = make_opt("c3d")
    model = network.get_network(opt)
    weights = synthetic.keyed_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(weights)
    oracle = R.oracle_network(opt)
    oracle.load_state_dict(weights)
    n = 2
    for m in (model, oracle):
        assert isinstance(m.dropout, torch.nn.Dropout) and m.dropout.p == 0.10
        m.dropout = _SharedMaskDropout(0.10, (n, 4096), seed=5)
    x = synthetic.synthetic_clips(n, 16, 112)
    _, z = synthetic.synthetic_targets(n)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    oracle.train()
    y_ref = R.embed(oracle, x)
    loss_ref = F.mse_loss(y_ref, z)
    loss_ref.backward()
    model.to(DEV).train()
    y = model(x.to(DEV))
    loss = F.mse_loss(y, z.to(DEV))
    loss.backward()
    assert rel_err(y.detach().cpu().numpy(), y_ref.detach().numpy()) < 1e-4
    assert abs(loss.item() / loss_ref.item() - 1) < 1e-4
    ref = {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert sorted(ref) == sorted(got) and len(got) == 20
    worst = max((rel_l2(got[k].cpu().numpy(), ref[k].numpy()), k) for k in ref)
    assert worst[0] < 2e-2, worst                   # fp32 vs fp32 through 8 ReLU / 5 max-pool layers (mask flips at ties)
