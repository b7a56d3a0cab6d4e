#!/usr/bin/env python
"""Standalone cost of the BatchNorm-backward sums in the temporal input-gradient epilogue:
zsv_conv3d_dgrad vs zsv_conv3d_dgrad_bnstats, and zsv_bn_bwd vs zsv_bn_bwd_from_stats, on the mid tensors of R(2+1)D-18."""
import ctypes
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from zeroshotvideoclassification_amd import _lib, ops  # noqa: E402

DEV = torch.device("cuda:0")
CASES = [("T1", (22, 144, 16, 56, 56), 64), ("T3", (22, 230, 8, 28, 28), 128), ("T4", (22, 288, 8, 28, 28), 128),
         ("T6", (22, 460, 4, 14, 14), 256), ("T7", (22, 576, 4, 14, 14), 256)]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    lib = _lib.load()
    for name, xs, cout in CASES:
        n, c, t, h, w = xs
        s = t * h * w
        x = torch.randn(*xs, device=DEV)
        gamma, beta = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.3
        wt = torch.randn(cout, c, 3, 1, 1, device=DEV) / (3 * c) ** 0.5
        dy = torch.randn(n, cout, t, h, w, device=DEV)
        d = ops.conv_desc(xs, wt.shape, 1, (1, 0, 0))
        tiles = lib.zsv_conv3d_dgrad_bnstat_tiles(ctypes.byref(d))
        pitch = (c + 15) // 16 * 16
        coef = torch.zeros((4, pitch), device=DEV)
        nb = lib.zsv_bn_workspace_bytes(n, c, s)
        bws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=DEV)
        _lib.check(lib.zsv_bn_fwd_train_coeffs(x.data_ptr(), n, c, s, gamma.data_ptr(), beta.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(),
                                               None, None, 0.1, 1e-5, None, 0, coef.data_ptr(), pitch, bws.data_ptr(), nb, None), "coeffs")
        nd = lib.zsv_conv3d_dgrad_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(int(nd), 16), dtype=torch.uint8, device=DEV)
        dx, bdx = torch.empty(xs, device=DEV), torch.empty(xs, device=DEV)
        dgamma, dbeta = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
        part = torch.empty((2, c, max(tiles, 1)), device=DEV)
        bn = _lib.BnBwdStats(x.data_ptr(), coef.data_ptr(), pitch, tiles, part.data_ptr())
        t_plain = timed(lambda: _lib.check(lib.zsv_conv3d_dgrad(ctypes.byref(d), dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), ws.data_ptr(), nd, None), "dgrad"))
        t_bwd = timed(lambda: _lib.check(lib.zsv_bn_bwd(dx.data_ptr(), x.data_ptr(), None, n, c, s, gamma.data_ptr(), beta.data_ptr(), coef[2].data_ptr(),
                                                        coef[3].data_ptr(), 2, bdx.data_ptr(), None, dgamma.data_ptr(), dbeta.data_ptr(), bws.data_ptr(), nb, None), "bn_bwd"))
        if tiles > 0:
            t_stats = timed(lambda: _lib.check(lib.zsv_conv3d_dgrad_bnstats(ctypes.byref(d), dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), ctypes.byref(bn),
                                                                            ws.data_ptr(), nd, None, None, 0), "dgrad_bnstats"))
            t_from = timed(lambda: _lib.check(lib.zsv_bn_bwd_from_stats(dx.data_ptr(), x.data_ptr(), n, c, s, gamma.data_ptr(), beta.data_ptr(), coef[2].data_ptr(),
                                                                        coef[3].data_ptr(), part.data_ptr(), tiles, bdx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                                        bws.data_ptr(), nb, None), "from_stats"))
            print(f"{name}: dgrad {t_plain:.3f} ms  + sums {t_stats:.3f} ms | bn_bwd {t_bwd:.3f} ms  from partials {t_from:.3f} ms | "
                  f"pair {t_plain + t_bwd:.3f} -> {t_stats + t_from:.3f} ms  (tiles {tiles})")
        else:
            print(f"{name}: dgrad {t_plain:.3f} ms | bn_bwd {t_bwd:.3f} ms (no epilogue for this geometry)")


if __name__ == "__main__":
    main()
