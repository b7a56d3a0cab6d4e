"""Does the mixed-precision (bf16) step train like the fp32 step at the headline size?  R(2+1)D-18, 22 clips per step, eight
distinct synthetic batches cycled for --steps steps from the same initial weights, Adam lr 1e-3: the two loss curves (every tenth
step) and clips/s.  (round 4 record)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace

from zeroshotvideoclassification_amd import network, optim, synthetic, train

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--batch", type=int, default=22)
ap.add_argument("--network", default="r2plus1d_18")
args = ap.parse_args()
dev = torch.device("cuda")
batches = []
for i in range(8):
    x = synthetic.synthetic_clips(args.batch, 16, 112, seed=500 + i).to(dev)
    _, z = synthetic.synthetic_targets(args.batch, rank=i)
    batches.append((x, z.to(dev)))
out = {"workload": f"{args.network}, {args.batch} clips 3x16x112x112 per step, 8 synthetic batches cycled, {args.steps} steps, Adam lr 1e-3, same initial weights"}
for mode in ("fp32", "bf16"):
    model = network.get_network(SimpleNamespace(network=args.network, fixconvs=False, nopretrained=False))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0))
    model.to(dev).train()
    crit = torch.nn.MSELoss()
    if mode == "bf16":
        opt = optim.FusedAdam(model.parameters(), lr=1e-3)
        scaler = optim.LossScaler()
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        scaler = None
    pacer = train.StepPacer(2)
    losses = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        x, z = batches[s % len(batches)]
        _, loss = train.train_step(model, opt, crit, x, z, scaler=scaler, pacer=pacer, autocast=(mode == "bf16"))
        if s % 10 == 0 or s == args.steps - 1:
            losses.append((s, loss))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[mode] = {"clips_per_s": round(args.batch * args.steps / dt, 1), "ms_per_step": round(1e3 * dt / args.steps, 3),
                 "loss": [[s, float(l.item())] for s, l in losses],
                 "loss_scale_end": float(scaler.get_scale()) if scaler is not None else None,
                 "nonfinite_parameters": int(sum((~torch.isfinite(p)).sum().item() for p in model.parameters()))}
l32, l16 = dict(out["fp32"]["loss"]), dict(out["bf16"]["loss"])
out["max_rel_loss_gap_after_step_20"] = round(max(abs(l16[s] - l32[s]) / l32[s] for s in l32 if s >= 20), 4)
print(json.dumps(out))
