"""Round 4: first contact of the bf16 training path (amp.py) with the GPU: agreement with the fp32 HIP path on the same
weights / clips (embeddings, loss, every gradient: cosine + relative L2), then step times at the headline size."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from types import SimpleNamespace

from zeroshotvideoclassification_amd import amp, network, ops, synthetic, train

dev = torch.device("cuda")


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def compare(net, n, frames, size):
    model = network.get_network(SimpleNamespace(network=net, fixconvs=False, nopretrained=False))
    sd = synthetic.keyed_state_dict(model.state_dict(), seed=0, bn_jitter=True)
    model.load_state_dict(sd)
    model.to(dev).train()
    x = synthetic.synthetic_clips(n, frames, size).to(dev)
    _, z = synthetic.synthetic_targets(n)
    z = z.to(dev)
    model.zero_grad(set_to_none=True)
    y32 = train.embed(model, x)
    l32 = F.mse_loss(y32, z)
    l32.backward()
    ops.join_wgrad_streams()
    g32 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    rm32 = {k: v.clone() for k, v in model.state_dict().items() if "running" in k}
    model.load_state_dict(sd)
    model.zero_grad(set_to_none=True)
    with amp.autocast():
        y16 = train.embed(model, x)
    l16 = F.mse_loss(y16, z)
    l16.backward()
    ops.join_wgrad_streams()
    torch.cuda.synchronize()
    g16 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    print(f"[{net} n={n} T={frames} {size}px] emb cos {cos(y32, y16):.6f} max|d| {float((y32 - y16).abs().max()):.4e}  loss {l32.item():.6e} vs {l16.item():.6e}")
    assert sorted(g32) == sorted(g16), (sorted(set(g32) ^ set(g16)))
    worst = []
    for k in g32:
        c = cos(g32[k], g16[k])
        rel = float((g32[k] - g16[k]).norm() / (g32[k].norm() + 1e-30))
        worst.append((c, rel, k))
    worst.sort()
    for c, rel, k in worst[:8]:
        print(f"    grad cos {c:.5f} rel-L2 {rel:.3e}  {k}")
    print(f"    median grad cos {sorted(w[0] for w in worst)[len(worst) // 2]:.5f}   min {worst[0][0]:.5f}")
    rm16 = {k: v for k, v in model.state_dict().items() if "running" in k}
    print("    running stats max rel diff", max(float((rm32[k] - rm16[k]).abs().max() / (rm32[k].abs().max() + 1e-12)) for k in rm32))
    return model


def timing(net, n=22, frames=16, size=112, steps=8):
    model = network.get_network(SimpleNamespace(network=net, fixconvs=False, nopretrained=False))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0))
    model.to(dev).train()
    x = synthetic.synthetic_clips(n, frames, size).to(dev)
    _, z = synthetic.synthetic_targets(n)
    z = z.to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = torch.nn.MSELoss()
    pacer = train.StepPacer(2)

    def step16():
        pacer.wait()
        opt.zero_grad(set_to_none=True)
        with amp.autocast():
            y = train.embed(model, x)
        loss = crit(y, z)
        loss.backward()
        ops.join_wgrad_streams()
        opt.step()
        pacer.mark()
        return loss

    for name, fn in (("bf16", step16), ("fp32", lambda: train.train_step(model, opt, crit, x, z, pacer=pacer)[1])):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"[{net} N={n}] {name} step {1e3 * dt:.2f} ms = {n / dt:.1f} clips/s  loss {loss.item():.4e}")
    # forward-only / forward+backward of the bf16 path
    def fwd():
        with torch.no_grad(), amp.autocast():
            model.train()
            train.embed(model, x)
    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fwd()
    torch.cuda.synchronize()
    print(f"[{net} N={n}] bf16 train-mode forward only {1e3 * (time.perf_counter() - t0) / steps:.2f} ms")


if __name__ == "__main__":
    compare("r2plus1d_18", 3, 8, 56)
    compare("r3d_18", 2, 8, 56)
    compare("r2plus1d_18", 2, 16, 112)
    if "--time" in sys.argv:
        timing("r2plus1d_18")
