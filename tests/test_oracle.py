"""The CPU oracle against (a) the committed golden vectors generated from the reference and
(b), in the build container only, the imported reference itself."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import case_inputs, load_golden, rel_err
from oracle import restatement as R
from oracle.reference_import import import_reference, reference_available
from zeroshotvideoclassification_amd import synthetic

torch.set_num_threads(8)


def _build(case):
    g = load_golden(case)
    model = R.oracle_network(R.make_opt(str(g["meta_network"])))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0, bn_jitter=bool(g["meta_bn_jitter"])))
    return g, model


@pytest.mark.parametrize("case", ["r2plus1d_small", "r3d_small"])
def test_oracle_reproduces_golden_forward_backward(case):
    g, model = _build(case)
    x, z = case_inputs(g)
    model.train()
    y = R.embed(model, x)
    loss = F.mse_loss(y, z)
    loss.backward()
    # generated on this torch build: bit-equal here; 1e-5 leaves room for another host's oneDNN path
    assert rel_err(y.detach().numpy(), g["emb_f32"]) < 1e-5
    assert abs(loss.item() / float(g["loss_f32"]) - 1) < 1e-5
    assert [k for k, p in model.named_parameters() if p.grad is None] == [str(k) for k in g["dead_params"]]
    params = dict(model.named_parameters())
    for name, norm in zip(g["grad_names"], g["grad_norm_f64"]):
        got = params[str(name)].grad.double().norm().item()
        assert abs(got - norm) <= 3e-2 * norm + 1e-12, str(name)          # fp32 vs fp64 noise floor


def test_oracle_c3d_eval_matches_golden():
    g, model = _build("c3d_eval")
    x, _ = case_inputs(g)
    model.eval()
    with torch.no_grad():
        y = R.embed(model, x)
    assert rel_err(y.numpy(), g["emb_eval_f32"]) < 1e-5


def test_oracle_accuracy_protocol():
    classes = synthetic.class_table(20, seed=1)
    pred = classes[[3, 4, 5, 6]] + 0.01 * torch.randn(4, 300, generator=torch.Generator().manual_seed(0))
    top1, top5 = R.compute_accuracy(pred, classes, classes[[3, 4, 5, 7]])
    assert top1 == 75.0 and top5 >= 75.0


@pytest.mark.skipif(not reference_available(), reason="/root/reference exists only in the build container")
@pytest.mark.parametrize("name", ["r2plus1d_18", "r3d_18", "c3d"])
def test_restatement_is_pinned_to_the_imported_reference(name):
    ref_network, _ = import_reference()
    opt = R.make_opt(name)
    ref, mine = ref_network.get_network(opt), R.oracle_network(opt)
    assert list(ref.state_dict().keys()) == list(mine.state_dict().keys())
    weights = synthetic.keyed_state_dict(ref.state_dict(), seed=5, bn_jitter=True)
    ref.load_state_dict(weights)
    mine.load_state_dict(weights)
    size = 112 if name == "c3d" else 32
    frames = 16 if name == "c3d" else 4
    x = synthetic.synthetic_clips(1 if name == "c3d" else 2, frames, size, seed=8)
    for m in (ref, mine):
        m.eval() if name == "c3d" else m.train()
    ya, yb = R.embed(ref, x), R.embed(mine, x)
    assert torch.equal(ya, yb)
    ya.sum().backward()
    yb.sum().backward()
    for (k, p), (_, q) in zip(ref.named_parameters(), mine.named_parameters()):
        assert (p.grad is None) == (q.grad is None), k
        if p.grad is not None:
            assert torch.equal(p.grad, q.grad), k


@pytest.mark.skipif(not reference_available(), reason="/root/reference exists only in the build container")
@pytest.mark.parametrize("order", ["transforms_first", "network_first"])
def test_reference_importers_work_in_either_order(order):
    """oracle/reference_import.py: the torchvision stand-in installed by one importer must not break the other (a stub
    ``torchvision`` without its sub-modules used to satisfy the second importer's ``import torchvision``)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    calls = ["t = RI.import_reference_transforms(); assert t.get_transform(True)", "n, r = RI.import_reference(); assert n.get_network and r.r2plus1d_18"]
    if order == "network_first":
        calls.reverse()
    code = "from oracle import reference_import as RI; " + "; ".join(calls) + "; print('ok')"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-800:]


def test_t32_batch_fixture_is_reproduced_by_the_oracle():
    """tests/golden/r2plus1d_t32_batch.npz (reference output, 4 clips x 32 frames, eval mode) against the restatement."""
    g = load_golden("r2plus1d_t32_batch")
    model = R.oracle_network(R.make_opt(str(g["meta_network"])))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0, bn_jitter=bool(g["meta_bn_jitter"])))
    model.eval()
    x = synthetic.synthetic_clips(int(g["meta_n"]), int(g["meta_frames"]), int(g["meta_size"]), seed=int(g["meta_seed"]))
    with torch.no_grad():
        y = R.embed(model, x)
    assert rel_err(y.numpy(), g["emb_eval_t32_f32"]) < 1e-5
