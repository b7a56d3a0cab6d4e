#!/usr/bin/env python
"""Micro-benchmark of the convolution entry points on the R(2+1)D-18 layer shapes (HIP events).

    python tools/conv_bench.py [--n 22] [--shapes S1,T1] [--kinds fwd,dgrad,wgrad] [--iters 10]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zeroshotvideoclassification_amd import _lib, ops

# name: (Cin, Cout, k, s, p, T, H, W)  -- SURVEY section 2a
SHAPES = {
    "S0": (3, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3), 16, 112, 112),
    "T0": (45, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), 16, 56, 56),
    "S1": (64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), 16, 56, 56),
    "T1": (144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), 16, 56, 56),
    "S2": (64, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1), 16, 56, 56),
    "T2": (230, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0), 16, 28, 28),
    "S3": (128, 230, (1, 3, 3), (1, 1, 1), (0, 1, 1), 8, 28, 28),
    "T3": (230, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), 8, 28, 28),
    "S4": (128, 288, (1, 3, 3), (1, 1, 1), (0, 1, 1), 8, 28, 28),
    "T4": (288, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), 8, 28, 28),
    "S5": (128, 460, (1, 3, 3), (1, 2, 2), (0, 1, 1), 8, 28, 28),
    "T5": (460, 256, (3, 1, 1), (2, 1, 1), (1, 0, 0), 8, 14, 14),
    "S6": (256, 460, (1, 3, 3), (1, 1, 1), (0, 1, 1), 4, 14, 14),
    "T6": (460, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), 4, 14, 14),
    "S7": (256, 576, (1, 3, 3), (1, 1, 1), (0, 1, 1), 4, 14, 14),
    "T7": (576, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), 4, 14, 14),
    "S8": (256, 921, (1, 3, 3), (1, 2, 2), (0, 1, 1), 4, 14, 14),
    "T8": (921, 512, (3, 1, 1), (2, 1, 1), (1, 0, 0), 4, 7, 7),
    "S9": (512, 921, (1, 3, 3), (1, 1, 1), (0, 1, 1), 2, 7, 7),
    "T9": (921, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), 2, 7, 7),
    "S10": (512, 1152, (1, 3, 3), (1, 1, 1), (0, 1, 1), 2, 7, 7),
    "T10": (1152, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), 2, 7, 7),
    "P1": (64, 128, (1, 1, 1), (2, 2, 2), (0, 0, 0), 16, 56, 56),
    "P2": (128, 256, (1, 1, 1), (2, 2, 2), (0, 0, 0), 8, 28, 28),
    "P3": (256, 512, (1, 1, 1), (2, 2, 2), (0, 0, 0), 4, 14, 14),
    # C3D (network.py:102-117): 3x3x3, stride 1, pad 1 -- not part of --shapes all
    "C2": (64, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), 16, 56, 56),
    "C3a": (128, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1), 8, 28, 28),
    "C3b": (256, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1), 8, 28, 28),
    "C4a": (256, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1), 4, 14, 14),
    "C4b": (512, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1), 4, 14, 14),
    "C5": (512, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1), 2, 7, 7),
    # R3D-18 / MC3-18 (resnet.py:23-30): 3x3x3 stride 1 -- not part of --shapes all
    "R1": (64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), 16, 56, 56),
    "R2": (128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), 8, 28, 28),
    "R3": (256, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1), 4, 14, 14),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=22)
    ap.add_argument("--shapes", default="S1,T1")
    ap.add_argument("--kinds", default="fwd,dgrad,wgrad")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--pre", action="store_true",
                    help="forward / wgrad through the BatchNorm-folded entry points (zsv_conv3d_fwd_pre / _wgrad_pre: the kernels a "
                         "training step runs for the stride-1 temporal convolutions) where the geometry supports them")
    ap.add_argument("--stats", action="store_true", help="forward with the BatchNorm partial statistics in the epilogue (the training forward) where supported")
    ap.add_argument("--add", action="store_true", help="dgrad with the fused shortcut-gradient add where supported")
    ap.add_argument("--markers", action="store_true",
                    help="launch a one-element fill kernel in front of every (shape, kind) group: tools/kernel_breakdown.py splits a "
                         "rocprofv3 kernel trace of this run at those markers")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    names = [n for n in SHAPES if n[0] not in "CR"] if args.shapes == "all" else args.shapes.split(",")
    tot = {}
    for name in names:
        cin, cout, k, s, p, t, h, w = SHAPES[name]
        x = torch.randn(args.n, cin, t, h, w, device=dev)
        wt = torch.randn(cout, cin, *k, device=dev) * 0.05
        d = ops.conv_desc(x.shape, wt.shape, s, p)
        y = torch.empty(d.N, d.Cout, d.To, d.Ho, d.Wo, device=dev)
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.empty_like(wt)
        nb = lib.zsv_conv3d_wgrad_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        flops = 2.0 * d.N * d.Cout * d.To * d.Ho * d.Wo * cin * k[0] * k[1] * k[2]
        nf = lib.zsv_conv3d_fwd_workspace_bytes(ctypes.byref(d))
        nd = lib.zsv_conv3d_dgrad_workspace_bytes(ctypes.byref(d))
        wsf = torch.empty(max(nf, nd, 16), dtype=torch.uint8, device=dev)
        calls = {
            "fwd": lambda: lib.zsv_conv3d_fwd(ctypes.byref(d), x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), 0, wsf.data_ptr(), nf, stream),
            "dgrad": lambda: lib.zsv_conv3d_dgrad(ctypes.byref(d), dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), wsf.data_ptr(), nd, stream),
            "wgrad": lambda: lib.zsv_conv3d_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, stream),
        }
        if args.stats:
            tiles = lib.zsv_conv3d_fwd_stat_tiles(ctypes.byref(d), y.data_ptr())
            if tiles > 0:
                st = torch.empty(2, cout, tiles, device=dev)
                calls["fwd"] = lambda: lib.zsv_conv3d_fwd_stats(ctypes.byref(d), x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), 0, st.data_ptr(), tiles, wsf.data_ptr(), nf, stream)
        if args.add and lib.zsv_conv3d_dgrad_add_supported(ctypes.byref(d)):
            addt = torch.randn_like(x)
            calls["dgrad"] = lambda: lib.zsv_conv3d_dgrad_add(ctypes.byref(d), dy.data_ptr(), wt.data_ptr(), addt.data_ptr(), dx.data_ptr(), wsf.data_ptr(), nd, stream)
        if args.pre and lib.zsv_conv3d_pre_supported(ctypes.byref(d)):
            pitch = (cin + 15) // 16 * 16
            coef = torch.zeros(2, pitch, device=dev)
            coef[0, :cin] = 1.0 + 0.1 * torch.randn(cin, device=dev)
            coef[1, :cin] = 0.1 * torch.randn(cin, device=dev)
            tiles_p = lib.zsv_conv3d_fwd_stat_tiles(ctypes.byref(d), y.data_ptr()) if args.stats else 0
            st_p = torch.empty(2, cout, max(tiles_p, 1), device=dev)
            calls["fwd"] = lambda: lib.zsv_conv3d_fwd_pre(ctypes.byref(d), x.data_ptr(), coef.data_ptr(), pitch, wt.data_ptr(), y.data_ptr(), st_p.data_ptr() if tiles_p > 0 else None, tiles_p, wsf.data_ptr(), nf, stream)
            calls["wgrad"] = lambda: lib.zsv_conv3d_wgrad_pre(ctypes.byref(d), x.data_ptr(), coef.data_ptr(), pitch, dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nb, stream)
        for kind in args.kinds.split(","):
            fn = calls[kind]
            if args.markers:
                print(f"MARK {name} {kind} {flops:.0f}", flush=True)
                torch.empty(1, device=dev).fill_(1.0)
            for _ in range(2):
                assert fn() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            tot[kind] = tot.get(kind, 0.0) + ms
            print(f"{name:4s} {kind:6s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s  "
                  f"(M={cout if kind != 'dgrad' else cin} K={cin * k[0] * k[1] * k[2]} P={d.N * d.To * d.Ho * d.Wo})", flush=True)
    print("total ms:", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
