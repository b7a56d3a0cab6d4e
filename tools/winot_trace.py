#!/usr/bin/env python
"""Where a workgroup of the temporal F(4,3) kernel spends its life: run with the trace build of the library
    tools/variant.sh trace conv_wino.hip -DZSV_WINOT_TRACE
    ZSV_LIB_PATH=build/variants/libzsv_trace.so python tools/winot_trace.py [T1|T3|...] [fwd|dgrad]
Every workgroup's wave 0 leaves s_memtime at: entry, before the first DMA, first chunk landed, chunk loop done, stores issued,
stores acknowledged (+ its XCC / CU).  Prints the phase lengths (in s_memtime ticks and microseconds of the whole launch)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from zeroshotvideoclassification_amd import _lib, ops  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = {"T1": ((22, 144, 16, 56, 56), 64), "T3": ((22, 230, 8, 28, 28), 128), "T4": ((22, 288, 8, 28, 28), 128),
          "T6": ((22, 460, 4, 14, 14), 256),
          # spatial 1x3x3 layers (conv_wino4_kernel): (input shape, output channels)
          "S1": ((22, 64, 16, 56, 56), 144), "S3": ((22, 128, 8, 28, 28), 230), "S4": ((22, 128, 8, 28, 28), 288)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "T1"
    kind = sys.argv[2] if len(sys.argv) > 2 else "dgrad"
    xs, cout = SHAPES[name]
    n, c, t, h, w = xs
    lib = _lib.load()
    spatial = name.startswith("S")
    wt = torch.randn(*((cout, c, 1, 3, 3) if spatial else (cout, c, 3, 1, 1)), device=DEV) / (3 * c) ** 0.5
    d = ops.conv_desc(xs, wt.shape, 1, (0, 1, 1) if spatial else (1, 0, 0))
    ys = (n, cout, t, h, w)
    global NCHUNKS
    NCHUNKS = ((cout if kind == "dgrad" else c) + 15) // 16 * (3 if spatial else 1)
    src_shape, out_shape = (ys, xs) if kind == "dgrad" else (xs, ys)
    src = torch.randn(*src_shape, device=DEV)
    out_elems = int(np.prod(out_shape))
    max_wgs = 1 << 16
    out = torch.zeros(out_elems + max_wgs * 24, device=DEV)               # room for 12 x u64 per workgroup behind the tensor
    if kind == "dgrad":
        nb = lib.zsv_conv3d_dgrad_workspace_bytes(ctypes.byref(d))
    else:
        nb = lib.zsv_conv3d_fwd_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=DEV)

    def run():
        if kind == "dgrad":
            _lib.check(lib.zsv_conv3d_dgrad(ctypes.byref(d), src.data_ptr(), wt.data_ptr(), out.data_ptr(), ws.data_ptr(), nb, None), "dgrad")
        else:
            _lib.check(lib.zsv_conv3d_fwd(ctypes.byref(d), src.data_ptr(), wt.data_ptr(), None, out.data_ptr(), 0, ws.data_ptr(), nb, None), "fwd")

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    out[out_elems:].zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    rec = out[out_elems:].cpu().numpy().view(np.uint64).reshape(-1, 12)
    rec = rec[rec[:, 0] != 0]
    if not len(rec):
        raise SystemExit("no trace records: is ZSV_LIB_PATH the trace build, and does this geometry run conv_winot4_kernel?")
    tt = rec[:, :6].astype(np.int64)
    hw = rec[:, 6]
    # the counters of the eight XCDs are not aligned with each other (and XCC_ID reads 0 here): workgroups are grouped by
    # clusters of their entry times, spans are taken per cluster
    order = np.argsort(tt[:, 0])
    gaps = np.diff(tt[order, 0])
    cuts = np.where(gaps > 20 * max(np.median(gaps), 1) + 100000)[0]
    xcc = np.zeros(len(tt), dtype=np.int64)
    for j, cpos in enumerate(cuts):
        xcc[order[cpos + 1:]] = j + 1
    spans = [tt[xcc == x, 5].max() - tt[xcc == x, 0].min() for x in np.unique(xcc)]
    print(f"  {len(spans)} counter groups, spans {sorted(int(v) for v in spans)}")
    span = float(np.median(spans))
    tick_us = ms * 1e3 / span                                           # (launch overhead included: a slight over-estimate)
    print(f"{name} {kind}: {len(rec)} workgroups, launch {ms * 1e3:.1f} us, span {span:.0f} ticks -> {tick_us * 1e3:.2f} ns per tick")
    names = ["entry -> first DMA issued (arguments, addresses)", "first chunk landed", "chunk loop", "output transform + stores issued",
             "stores acknowledged"]
    ph = np.diff(tt, axis=1)
    life = tt[:, 5] - tt[:, 0]
    for i, nm in enumerate(names):
        v = ph[:, i] * tick_us
        print(f"  {nm:52s} mean {v.mean():7.2f} us   p10 {np.percentile(v, 10):7.2f}   p50 {np.percentile(v, 50):7.2f}   p90 {np.percentile(v, 90):7.2f}"
              f"   {100 * ph[:, i].sum() / life.sum():5.1f} % of a workgroup's life")
    print(f"  workgroup life mean {life.mean() * tick_us:.2f} us; workgroups x life / span = {life.sum() / span:.1f} resident on average "
          f"(512 slots)")
    laps = rec[:, 8:11].astype(np.int64)
    loop = ph[:, 2].astype(np.float64)
    for i, nm in enumerate(["before the first MFMA of a chunk (DMA issue, U fragments, first V)", "the four k steps (V transforms + MFMAs)",
                            "chunk-end wait (next chunk's DMAs + barrier)"]):
        print(f"    inside the chunk loop: {nm:70s} {100 * laps[:, i].sum() / loop.sum():5.1f} %   {laps[:, i].mean() * tick_us / max(1, NCHUNKS):6.2f} us per chunk")
    extra = rec[:, 11]
    l3 = (extra >> np.uint64(32)).astype(np.int64)
    l4 = (extra & np.uint64(0xFFFFFFFF)).astype(np.int64)
    print(f"      of the first: DMA instructions of the next chunk {l3.mean() * tick_us / max(1, NCHUNKS):6.2f} us per chunk, "
          f"U fragments + first image values landed {l4.mean() * tick_us / max(1, NCHUNKS):6.2f} us per chunk")
    hwid = hw & np.uint64(0xFFFFFFFF)
    cu = (hwid >> np.uint64(8)) & np.uint64(0xF)
    se = (hwid >> np.uint64(13)) & np.uint64(0x7)
    key = (xcc * 8 + se.astype(np.int64)) * 16 + cu.astype(np.int64)
    print(f"  distinct (xcc, se, cu): {len(np.unique(key))}")
    # how the two workgroups of one CU sit against each other: for every workgroup, the share of ITS chunk loop during which
    # another workgroup of the same CU was also inside its chunk loop
    shares = []
    for k in np.unique(key)[:64]:
        idx = np.where(key == k)[0]
        s, e = tt[idx, 2], tt[idx, 3]
        for i in range(len(idx)):
            ov = np.clip(np.minimum(e, e[i]) - np.maximum(s, s[i]), 0, None)
            ov[i] = 0
            shares.append(ov.sum() / max(e[i] - s[i], 1))
    print(f"  share of a chunk loop spent next to another workgroup's chunk loop on the same CU: mean {np.mean(shares):.2f}")


if __name__ == "__main__":
    main()
