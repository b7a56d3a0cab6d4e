#!/usr/bin/env python
"""Micro-benchmark of the bf16 TRAINING kernels (amp.py) on the R(2+1)D-18 layer shapes, channels-last bf16 operands, HIP events:
forward (zsv_conv3d_bf16_fwd on the packed fp32 master weight), input gradient (the forward kernel on the swapped / flipped weight,
residue classes for strides: amp.Bf16TrainPath._dgrad, pack included), weight gradient (zsv_conv3d_bf16_wgrad, or the converted-operand
fp32 path where that kernel does not apply: amp.Bf16TrainPath._wgrad), BatchNorm forward / backward on the convolution's output.

    python tools/conv_bench_bf16.py [--n 22] [--shapes S1,T1] [--kinds fwd,dgrad,wgrad,bn_fwd,bn_bwd] [--iters 10]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from conv_bench import SHAPES
from zeroshotvideoclassification_amd import amp, inference, ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=22)
    ap.add_argument("--shapes", default="S1,T1")
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad,bn_fwd,bn_bwd")
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda")
    names = [n for n in SHAPES if n[0] not in "CR" and n != "S0"] if args.shapes == "all" else args.shapes.split(",")
    tot = {}
    for name in names:
        cin, cout, k, s, p, t, h, w = SHAPES[name]
        conv = torch.nn.Conv3d(cin, cout, k, stride=s, padding=p, bias=False).to(dev)
        bn = torch.nn.BatchNorm3d(cout).to(dev).train()
        u = amp._Unit(conv, bn, True)
        x = amp.ncdhw_to_cl_bf16(torch.randn(args.n, cin, t, h, w, device=dev))
        d = u.desc(args.n, t, h, w)
        blob = inference.pack_conv(d, conv.weight.detach(), None, None)
        z = inference.conv_bf16(d, x, blob, None, False)
        y, mean, invstd = amp.bn_cl_fwd_train(z, bn, None, True)
        dz = (torch.randn(z.shape, device=dev) * 0.1).to(torch.bfloat16)
        dz[..., cout:] = 0
        rec = amp._Record()
        rec.unit, rec.desc, rec.x, rec.clips = u, d, x, None
        flops = 2.0 * d.N * d.Cout * d.To * d.Ho * d.Wo * cin * k[0] * k[1] * k[2]

        def wgrad():
            amp.Bf16TrainPath._wgrad(rec, dz)
            ops.join_wgrad_streams()

        calls = {
            "fwd": lambda: inference.conv_bf16(d, x, blob, None, False),
            "dgrad": lambda: amp.Bf16TrainPath._dgrad(rec, dz),
            "wgrad": wgrad,
            "bn_fwd": lambda: amp.bn_cl_fwd_train(z, bn, None, True),
            "bn_bwd": lambda: amp.bn_cl_bwd(dz, y, z, bn, mean, invstd, True, False),
        }
        for kind in args.kinds.split(","):
            fn = calls[kind]
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            tot[kind] = tot.get(kind, 0.0) + ms
            if kind.startswith("bn"):
                nbytes = z.numel() * 2 * (3 if kind == "bn_fwd" else 7)          # fwd: z, z, y;  bwd: (dy, y, z) x 2 + dz
                print(f"{name:4s} {kind:6s} {ms:8.3f} ms  {nbytes / ms / 1e9:7.2f} TB/s of algorithmic traffic ({nbytes / 1e6:.0f} MB)", flush=True)
            else:
                print(f"{name:4s} {kind:6s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s  (Cin={cin} Cout={cout} P={d.N * d.To * d.Ho * d.Wo})", flush=True)
    print("total ms:", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
