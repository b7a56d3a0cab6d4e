"""Not a test: times the oracle's plain-PyTorch restatement (torch.nn.Conv3d -> MIOpen, torch BatchNorm,
autograd, torch.optim.Adam) for the benchmark step on the same GPU, as a vendor-library reference point
next to bench.py.  Lives under tests/ because only tests/ may import oracle/.

    python tests/torch_gpu_reference_timing.py [--batch 22] [--steps 5]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restatement as R  # noqa: E402
from zeroshotvideoclassification_amd import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=22)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--benchmark", action="store_true", help="torch.backends.cudnn.benchmark = True (MIOpen searches its solvers)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = bool(a.benchmark)
    model = R.oracle_network(R.make_opt("r2plus1d_18"))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0))
    model.to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x = synthetic.synthetic_clips(a.batch, 16, 112).to(dev)
    _, z = synthetic.synthetic_targets(a.batch)
    z = z.to(dev)
    for i in range(a.warmup):
        R.train_step(model, opt, x, z)
        torch.cuda.synchronize()
        print(f"warm-up {i} done", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        R.train_step(model, opt, x, z)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"what": "plain PyTorch (MIOpen / ATen) training step on the same GPU, R(2+1)D-18 fp32",
                      "cudnn_benchmark": bool(a.benchmark), "batch": a.batch, "ms_per_step": round(dt * 1e3, 2), "clips_per_s": round(a.batch / dt, 1),
                      "torch": torch.__version__}))


if __name__ == "__main__":
    main()
