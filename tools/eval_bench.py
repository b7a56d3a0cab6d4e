#!/usr/bin/env python
"""Config E geometry: the reference's `evaluate()` protocol (main.py:224-325) on
32-frame clips -- eval-mode forward under no_grad + cosine nearest-class search against the three
"kinetics2others" class tables (UCF101 101, HMDB51 51, ActivityNet 200 classes; dataset.py:34-90),
synthetic clips and unit-norm class tables.

    python tools/eval_bench.py [--batch 22] [--frames 32] [--batches 6] [--dtype fp32|bf16]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace
from zeroshotvideoclassification_amd import network, synthetic, train


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=22)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp32-folded", "bf16"])
    args = ap.parse_args()
    dev = torch.device("cuda")
    model = network.get_network(SimpleNamespace(network="r2plus1d_18", fixconvs=False, nopretrained=False))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0, bn_jitter=True))
    model.to(dev).eval()
    results = {}
    for name, ncls in (("ucf101", 101), ("hmdb51", 51), ("activitynet", 200)):
        table = synthetic.class_table(ncls, seed=1000 + ncls)
        batches = []
        for i in range(args.batches):
            x = synthetic.synthetic_clips(args.batch, args.frames, 112, seed=7000 + i).to(dev)
            labels, z = synthetic.synthetic_targets(args.batch, ncls, seed=1000 + ncls, rank=i)
            batches.append((x, labels, z))
        dt_ = {"bf16": torch.bfloat16, "fp32-folded": torch.float32}.get(args.dtype)
        train.evaluate(model, batches[:1], table, device=dev, splits=0, dtype=dt_)          # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = train.evaluate(model, batches, table, device=dev, dtype=dt_)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res["clips_per_s"] = round(res["n"] / dt, 1)
        results[name] = res
    print(json.dumps({"metric": f"eval clips/s, R(2+1)D-18, 32x112x112, {args.dtype}, eval-mode BN + cosine NN",
                      "batch": args.batch, "frames": args.frames, "results": results}))


if __name__ == "__main__":
    main()
