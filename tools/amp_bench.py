"""bf16 training step (amp.autocast) at the headline size: step time, device time of the phases, for rocprofv3 runs.
    python tools/amp_bench.py [--steps 8] [--batch 22] [--optimizer torch|fused]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace

from zeroshotvideoclassification_amd import amp, network, ops, optim, synthetic, train

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--batch", type=int, default=22)
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--size", type=int, default=112)
ap.add_argument("--network", default="r2plus1d_18")
ap.add_argument("--fp32", action="store_true", help="time the fp32 step instead (same loop)")
ap.add_argument("--optimizer", choices=["torch", "fused"], default="torch")
ap.add_argument("--graph", action="store_true", help="the bf16 trunk as two hipGraphs (amp.autocast(graph=True))")
args = ap.parse_args()
dev = torch.device("cuda")
model = network.get_network(SimpleNamespace(network=args.network, fixconvs=False, nopretrained=False))
model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0))
model.to(dev).train()
x = synthetic.synthetic_clips(args.batch, args.frames, args.size).to(dev)
_, z = synthetic.synthetic_targets(args.batch)
z = z.to(dev)
opt = optim.FusedAdam(model.parameters(), lr=1e-3) if args.optimizer == "fused" else torch.optim.Adam(model.parameters(), lr=1e-3)
crit = torch.nn.MSELoss()
pacer = train.StepPacer(2)


def step():
    return train.train_step(model, opt, crit, x, z, pacer=pacer, autocast=not args.fp32, graph=args.graph)[1]


for _ in range(3):
    step()
torch.cuda.synchronize()
marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
t0 = time.perf_counter()
marks[0].record()
host = []
for i in range(args.steps):
    loss = step()
    marks[i + 1].record()
    host.append(time.perf_counter())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
dev_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
print(f"{'fp32' if args.fp32 else 'bf16'} {args.network} N={args.batch}: {1e3 * dt:.2f} ms/step = {args.batch / dt:.1f} clips/s; "
      f"device median {dev_ms[len(dev_ms) // 2]:.2f} ms; host per step {1e3 * (host[-1] - t0) / args.steps:.2f} ms; loss {loss.item():.4e}")
# host cost of one step on an idle queue
torch.cuda.synchronize()
t = time.perf_counter()
step()
print(f"host enqueue on an idle queue: {1e3 * (time.perf_counter() - t):.2f} ms")
torch.cuda.synchronize()
