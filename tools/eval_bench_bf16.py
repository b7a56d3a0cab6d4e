"""Throughput of the bf16 eval forward (BASELINE config 5 geometry: 32-frame 112x112 clips).
python tools/eval_bench_bf16.py [--batch 22] [--frames 32] [--iters 10]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace

from zeroshotvideoclassification_amd import inference, network, synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=22)
    ap.add_argument("--frames", type=int, default=32, help="c3d: 16 (fc6 expects 512x1x4x4 features)")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--network", default="r2plus1d_18")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = network.get_network(SimpleNamespace(network=a.network, fixconvs=False, nopretrained=False))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0, bn_jitter=True))
    model = model.to(dev).eval()
    x = synthetic.synthetic_clips(a.batch, a.frames, 112).to(dev)
    eng = inference.engine_for(model, torch.bfloat16)        # Bf16Engine (VideoResNet trunks) or Bf16EngineC3D
    for _ in range(3):
        eng(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        eng(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    gflop = (77.06 if a.network == "c3d" else 81.04 * a.frames / 16)      # forward GFLOP per clip
    with torch.no_grad():
        for _ in range(2):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
        dt32 = (time.perf_counter() - t0) / 3
    print(json.dumps({"workload": f"{a.network} eval forward, {a.batch} clips 3x{a.frames}x112x112", "bf16_clips_per_s": a.batch / dt,
                      "bf16_ms": dt * 1e3, "bf16_tflops": a.batch * gflop / dt / 1e3,
                      "fp32_clips_per_s": a.batch / dt32, "fp32_ms": dt32 * 1e3}))


if __name__ == "__main__":
    main()
