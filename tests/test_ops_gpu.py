"""Per-op parity of the HIP kernels (through the C ABI) against torch CPU fp64.

Tolerances: fp32 accumulation over K terms against an fp64 reference; the bound used is
max|err| <= 2e-5 * max|ref| (+ tiny abs), far inside the 1e-3 bar north_star states.
"""
import numpy as np
import pytest
import zlib

import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from zeroshotvideoclassification_amd import ops  # noqa: E402

DEV = "cuda"


def close(a, ref, rtol=2e-5, what=""):
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert a.shape == ref.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(ref.shape)}"
    scale = ref.abs().max().item() + 1e-30
    err = (a - ref).abs().max().item()
    assert err <= rtol * scale + 1e-12, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e})"


def uses_f43(w):
    """F(4,3) along W: W % 4 == 0, or a row that nearly fills a power-of-two virtual row of 8 / 16 / 32 / 64 voxels
    (conv_wino.hip wino_vw_width: 7, 14, 15, 29-31, 61-63)."""
    if w % 4 == 0:
        return True
    for wv in (8, 16, 32, 64):
        if w < wv:
            return w > wv - 4 and 8 * w >= 7 * wv
    return False


def wino_vs_direct_tol(w):
    """The kw taps run in Winograd form: F(4,3) when `uses_f43` (transform constants up to 8: measured <= 6e-6 of the
    output range from the direct kernel, K up to 6912), F(2,3) otherwise (<= 2e-6)."""
    return 1e-5 if uses_f43(w) else 5e-6


CONV_CASES = [
    # name, N, Cin, T, H, W, Cout, k, s, p, bias
    ("spatial_s1", 2, 8, 3, 10, 12, 20, (1, 3, 3), (1, 1, 1), (0, 1, 1), False),
    ("spatial_s2", 2, 16, 2, 14, 14, 45, (1, 3, 3), (1, 2, 2), (0, 1, 1), False),
    ("spatial_144", 1, 64, 2, 12, 12, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1), False),
    ("stem_7x7", 2, 3, 2, 30, 30, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3), False),
    ("temporal_s1", 2, 45, 6, 7, 9, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), False),
    ("temporal_s2", 2, 30, 8, 6, 6, 17, (3, 1, 1), (2, 1, 1), (1, 0, 0), False),
    ("pointwise_s2", 2, 24, 4, 8, 8, 40, (1, 1, 1), (2, 2, 2), (0, 0, 0), False),
    ("full_333", 2, 5, 4, 9, 9, 12, (3, 3, 3), (1, 1, 1), (1, 1, 1), True),
    ("full_333_s2", 1, 12, 6, 11, 11, 20, (3, 3, 3), (2, 2, 2), (1, 1, 1), False),
    ("stem_3x7x7", 1, 3, 4, 20, 20, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), False),
    ("wide_230", 1, 64, 2, 8, 8, 230, (1, 3, 3), (1, 2, 2), (0, 1, 1), False),
    ("deep_k", 3, 288, 2, 7, 7, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), False),
    ("one_voxel", 1, 7, 1, 1, 1, 5, (1, 1, 1), (1, 1, 1), (0, 0, 0), True),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv3d_fwd_dgrad_wgrad(case):
    name, n, cin, t, h, w, cout, k, s, p, has_bias = case
    g = torch.Generator().manual_seed(zlib.crc32(str(name).encode()))    # (not hash(): randomised per process, a failure could not be replayed)
    x = torch.randn(n, cin, t, h, w, generator=g)
    wt = torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])
    b = torch.randn(cout, generator=g) if has_bias else None
    xr = x.double().requires_grad_()
    wr = wt.double().requires_grad_()
    br = b.double().requires_grad_() if has_bias else None
    yr = F.conv3d(xr, wr, br, stride=s, padding=p)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    xg = x.to(DEV).requires_grad_()
    wg = wt.to(DEV).requires_grad_()
    bg = b.to(DEV).requires_grad_() if has_bias else None
    yg = ops.conv3d(xg, wg, bg, s, p)
    close(yg, yr, what=f"{name} fwd")
    yg.backward(dy.to(DEV))
    close(xg.grad, xr.grad, what=f"{name} dgrad")
    close(wg.grad, wr.grad, what=f"{name} wgrad")
    if has_bias:
        close(bg.grad, br.grad, what=f"{name} dbias")


def test_conv3d_relu_fused_and_determinism():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 6, 4, 9, 9, generator=g)
    wt = torch.randn(10, 6, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(10, generator=g)
    xr, wr, br = x.double().requires_grad_(), wt.double().requires_grad_(), b.double().requires_grad_()
    yr = F.relu(F.conv3d(xr, wr, br, padding=1))
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    outs = []
    for _ in range(2):
        xg, wg, bg = x.to(DEV).requires_grad_(), wt.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        yg = ops.conv3d(xg, wg, bg, 1, 1, relu=True)
        yg.backward(dy.to(DEV))
        outs.append((yg.detach().clone(), xg.grad.clone(), wg.grad.clone(), bg.grad.clone()))
    close(outs[0][0], yr, what="conv+relu fwd")
    close(outs[0][1], xr.grad, what="conv+relu dgrad")
    close(outs[0][2], wr.grad, what="conv+relu wgrad")
    close(outs[0][3], br.grad, what="conv+relu dbias")
    for a, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a, b_), "kernels must be bitwise reproducible run to run"


def test_conv3d_wgrad_many_slices():
    # long voxel axis -> several slabs in the deterministic split reduction
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 8, 8, 28, 28, generator=g)
    wt = torch.randn(16, 8, 1, 3, 3, generator=g) * 0.1
    xr, wr = x.double(), wt.double().requires_grad_()
    yr = F.conv3d(xr, wr, padding=(0, 1, 1))
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xg, wg = x.to(DEV), wt.to(DEV).requires_grad_()
    ops.conv3d(xg, wg, None, 1, (0, 1, 1)).backward(dy.to(DEV))
    close(wg.grad, wr.grad, rtol=5e-5, what="wgrad slabs")


def test_conv3d_rejects_bad_inputs():
    x = torch.randn(1, 3, 2, 4, 4)
    w = torch.randn(4, 3, 1, 3, 3)
    with pytest.raises(RuntimeError):
        ops.conv3d(x, w)                      # CPU tensors: no fallback
    with pytest.raises(RuntimeError):
        ops.conv3d(x.to(DEV), torch.randn(4, 5, 1, 3, 3, device=DEV))
    with pytest.raises(RuntimeError):
        ops.conv3d(x.to(DEV).half(), w.to(DEV).half())


BN_CASES = [(2, 5, (3, 4, 4)), (3, 45, (2, 7, 7)), (2, 64, (4, 8, 8)), (4, 7, (1, 1, 1)), (2, 130, (2, 7, 7))]


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("use_res", [False, True])
@pytest.mark.parametrize("shape", BN_CASES, ids=[f"{n}x{c}x{'x'.join(map(str, s))}" for n, c, s in BN_CASES])
def test_batchnorm_train_fwd_bwd(shape, use_res, relu):
    n, c, sp = shape
    g = torch.Generator().manual_seed(n * 1000 + c)
    x = torch.randn(n, c, *sp, generator=g) * 2 + 0.5
    res = torch.randn(n, c, *sp, generator=g) if use_res else None
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.1
    rm0 = torch.randn(c, generator=g) * 0.1
    rv0 = torch.rand(c, generator=g) + 0.5
    dy = torch.randn(n, c, *sp, generator=g)

    xr = x.double().requires_grad_()
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    rr = res.double().requires_grad_() if use_res else None
    rm, rv = rm0.double().clone(), rv0.double().clone()
    yr = F.batch_norm(xr, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5)
    if use_res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    yr.backward(dy.double())

    xg = x.to(DEV).requires_grad_()
    gg, bg = gamma.to(DEV).requires_grad_(), beta.to(DEV).requires_grad_()
    rg = res.to(DEV).requires_grad_() if use_res else None
    rmg, rvg = rm0.to(DEV), rv0.to(DEV)
    yg = ops.batch_norm_act(xg, gg, bg, rmg, rvg, rg, True, 0.1, 1e-5, relu)
    close(yg, yr, what="bn fwd")
    close(rmg, rm, what="running_mean")
    close(rvg, rv, what="running_var")
    yg.backward(dy.to(DEV))
    close(xg.grad, xr.grad, rtol=1e-4, what="bn dx")
    close(gg.grad, gr.grad, rtol=1e-4, what="bn dgamma")
    close(bg.grad, br.grad, rtol=1e-4, what="bn dbeta")
    if use_res:
        close(rg.grad, rr.grad, what="bn dres")


def test_batchnorm_eval():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 9, 2, 5, 5, generator=g)
    gamma, beta = torch.rand(9, generator=g) + 0.5, torch.randn(9, generator=g)
    rm, rv = torch.randn(9, generator=g), torch.rand(9, generator=g) + 0.5
    ref = F.relu(F.batch_norm(x.double(), rm.double(), rv.double(), gamma.double(), beta.double(), training=False))
    out = ops.batch_norm_act(x.to(DEV), gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV), None, False, 0.1, 1e-5, True)
    close(out, ref, what="bn eval")


def test_batchnorm_large_mean_is_stable():
    # |mean| >> std: the sum/sumsq formulation must still give an accurate variance
    g = torch.Generator().manual_seed(8)
    x = torch.randn(4, 6, 4, 16, 16, generator=g) * 0.5 + 10.0
    ref = F.batch_norm(x.double(), None, None, None, None, training=True)
    out = ops.batch_norm_act(x.to(DEV), None, None, None, None, None, True, 0.1, 1e-5, False)
    close(out, ref, rtol=2e-4, what="bn shifted")


def test_batchnorm_mean_a_thousand_standard_deviations_away():
    """|mean| / std ~ 1e3 per channel (and a different offset per channel): E[x^2] - mean^2 in fp32 would lose
    the variance entirely; the shifted sums of bn_stats_kernel keep it.  Compared with the fp64 BatchNorm of the
    SAME fp32-rounded inputs, forward, running statistics and input gradient."""
    g = torch.Generator().manual_seed(11)
    c = 7
    offset = (torch.arange(c, dtype=torch.float32) - 3.0).view(1, c, 1, 1, 1) * 400.0 + 1000.0
    x = torch.randn(6, c, 4, 12, 12, generator=g) + offset
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    dy = torch.randn(x.shape, generator=g)
    xr = x.double().requires_grad_()
    rm, rv = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    ref = F.batch_norm(xr, rm, rv, gamma.double(), beta.double(), training=True, momentum=0.1, eps=1e-5)
    ref.backward(dy.double())
    xg = x.to(DEV).requires_grad_()
    rmg, rvg = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    out = ops.batch_norm_act(xg, gamma.to(DEV), beta.to(DEV), rmg, rvg, None, True, 0.1, 1e-5, False)
    # x - mean is exact to ~1e-4 absolute at |x| ~ 2e3 in fp32: that bounds the achievable accuracy of xhat
    close(out, ref, rtol=1e-3, what="bn far-from-zero mean")
    close(rvg, rv, rtol=1e-4, what="running_var far-from-zero mean")
    close(rmg, rm, rtol=1e-6, what="running_mean far-from-zero mean")
    out.backward(dy.to(DEV))
    close(xg.grad, xr.grad, rtol=2e-3, what="bn dx far-from-zero mean")


def test_relu_add_relu_meanpool():
    g = torch.Generator().manual_seed(2)
    for numel_shape in [(3, 5, 2, 7, 7), (1, 1, 1, 1, 3), (2, 4, 4, 8, 8)]:
        a = torch.randn(*numel_shape, generator=g)
        b = torch.randn(*numel_shape, generator=g)
        dy = torch.randn(*numel_shape, generator=g)
        ar, br = a.double().requires_grad_(), b.double().requires_grad_()
        yr = F.relu(ar + br)
        yr.backward(dy.double())
        ag, bg = a.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        yg = ops.add_relu(ag, bg)
        yg.backward(dy.to(DEV))
        close(yg, yr, what="add_relu")
        close(ag.grad, ar.grad, what="add_relu da")
        close(bg.grad, br.grad, what="add_relu db")

        ar2 = a.double().requires_grad_()
        F.relu(ar2).backward(dy.double())
        ag2 = a.to(DEV).requires_grad_()
        y2 = ops.relu(ag2)
        y2.backward(dy.to(DEV))
        close(y2, F.relu(a.double()), what="relu")
        close(ag2.grad, ar2.grad, what="relu bwd")

        ar3 = a.double().requires_grad_()
        mr = ar3.mean(dim=(2, 3, 4))
        dm = torch.randn(mr.shape, generator=g)
        mr.backward(dm.double())
        ag3 = a.to(DEV).requires_grad_()
        mg = ops.mean_pool(ag3)
        mg.backward(dm.to(DEV))
        close(mg, mr, what="meanpool")
        close(ag3.grad, ar3.grad, what="meanpool bwd")


POOL_CASES = [((1, 2, 2), (0, 0, 0), (2, 3, 4, 8, 8)), ((2, 2, 2), (0, 0, 0), (1, 5, 4, 6, 6)),
              ((2, 2, 2), (0, 1, 1), (2, 4, 2, 7, 7)), ((2, 2, 2), (0, 0, 0), (1, 2, 5, 9, 7)),
              ((2, 2, 2), (0, 1, 1), (2, 4, 2, 8, 8)), ((1, 2, 2), (0, 0, 0), (1, 3, 4, 12, 16)),      # W % 4 == 0: the quad backward
              ((2, 2, 2), (0, 0, 0), (3, 5, 4, 10, 12))]


@pytest.mark.parametrize("k,p,shape", POOL_CASES)
def test_maxpool3d(k, p, shape):
    g = torch.Generator().manual_seed(sum(shape))
    x = F.relu(torch.randn(*shape, generator=g))          # ties at 0 like post-ReLU activations
    xr = x.double().requires_grad_()
    yr = F.max_pool3d(xr, k, k, p)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xg = x.to(DEV).requires_grad_()
    yg = ops.max_pool3d(xg, k, k, p)
    yg.backward(dy.to(DEV))
    close(yg, yr, what="maxpool")
    close(xg.grad, xr.grad, what="maxpool bwd")


@pytest.mark.parametrize("rows,fin,fout,relu", [(22, 512, 512, True), (22, 512, 300, False), (3, 37, 11, False),
                                                 (1, 8192, 64, True)])
def test_linear(rows, fin, fout, relu):
    g = torch.Generator().manual_seed(rows + fin)
    x = torch.randn(rows, fin, generator=g)
    w = torch.randn(fout, fin, generator=g) / np.sqrt(fin)
    b = torch.randn(fout, generator=g)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    yr = F.linear(xr, wr, br)
    if relu:
        yr = F.relu(yr)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xg, wg, bg = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    yg = ops.linear(xg, wg, bg, relu=relu)
    yg.backward(dy.to(DEV))
    close(yg, yr, what="linear")
    close(xg.grad, xr.grad, what="linear dx")
    close(wg.grad, wr.grad, what="linear dw")
    close(bg.grad, br.grad, what="linear db")


def test_adam_step_matches_torch():
    g = torch.Generator().manual_seed(4)
    p0 = torch.randn(1000, generator=g)
    pr = p0.clone().requires_grad_()
    opt = torch.optim.Adam([pr], lr=1e-3)
    pg = p0.to(DEV)
    m = torch.zeros_like(pg)
    v = torch.zeros_like(pg)
    for step in range(1, 4):
        grad = torch.randn(1000, generator=g)
        pr.grad = grad.clone()
        opt.step()
        ops.adam_step_(pg, grad.to(DEV), m, v, step, 1e-3)
    close(pg, pr, rtol=1e-6, what="adam")


def test_conv3d_split_k_small_grid():
    # few output tiles + long reduction: the forward / dgrad kernels cut the K range into parts
    # (partial slabs + ordered reduce); layer4 geometry of R(2+1)D-18 scaled down in N
    g = torch.Generator().manual_seed(21)
    for (cin, cout, k, s, p, t, h, w) in [(256, 320, (1, 3, 3), (1, 1, 1), (0, 1, 1), 2, 7, 7),
                                          (320, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), 2, 7, 7),
                                          (128, 200, (1, 3, 3), (1, 2, 2), (0, 1, 1), 2, 14, 14)]:
        x = torch.randn(2, cin, t, h, w, generator=g)
        wt = torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])
        b = torch.randn(cout, generator=g)
        xr, wr, br = x.double().requires_grad_(), wt.double().requires_grad_(), b.double().requires_grad_()
        yr = F.relu(F.conv3d(xr, wr, br, stride=s, padding=p))
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy.double())
        xg, wg, bg = x.to(DEV).requires_grad_(), wt.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        yg = ops.conv3d(xg, wg, bg, s, p, relu=True)
        yg.backward(dy.to(DEV))
        close(yg, yr, what="split-K fwd")
        close(xg.grad, xr.grad, what="split-K dgrad")
        close(wg.grad, wr.grad, what="split-K wgrad")


def test_conv_epilogue_statistics_feed_batchnorm():
    # geometry on the fused path (voxel-contiguous output, S % 4 == 0): the BatchNorm that consumes
    # the convolution's epilogue partials must agree with the one that makes its own pass
    g = torch.Generator().manual_seed(33)
    x = torch.randn(3, 32, 4, 12, 12, generator=g).to(DEV)
    wt = (torch.randn(48, 32, 1, 3, 3, generator=g) * 0.1).to(DEV)
    gamma = (torch.rand(48, generator=g) + 0.5).to(DEV)
    beta = torch.randn(48, generator=g).to(DEV)
    y, stats = ops.conv3d(x, wt, None, 1, (0, 1, 1), want_stats=True)
    assert stats is not None and stats.shape[:2] == (2, 48)
    rm1, rv1 = torch.zeros(48, device=DEV), torch.ones(48, device=DEV)
    rm2, rv2 = torch.zeros(48, device=DEV), torch.ones(48, device=DEV)
    a = ops.batch_norm_act(y, gamma, beta, rm1, rv1, None, True, 0.1, 1e-5, True, stats=stats)
    b = ops.batch_norm_act(y, gamma, beta, rm2, rv2, None, True, 0.1, 1e-5, True)
    close(a, b, rtol=2e-6, what="bn(conv stats) vs bn(own stats)")
    close(rm1, rm2, rtol=1e-6, what="running mean")
    close(rv1, rv2, rtol=1e-6, what="running var")
    ref = F.relu(F.batch_norm(F.conv3d(x.double().cpu(), wt.double().cpu(), padding=(0, 1, 1)), None, None,
                              gamma.double().cpu(), beta.double().cpu(), training=True))
    close(a, ref, what="conv+bn+relu vs fp64")
    # geometries that cannot provide statistics return None and the BatchNorm makes its own pass
    _, none_stats = ops.conv3d(x, wt, None, (1, 2, 2), (0, 1, 1), want_stats=True)      # oS = 4*6*6 ok -> may fuse
    y2, s2 = ops.conv3d(x[:, :, :, :7, :7].contiguous(), wt, None, 1, (0, 1, 1), want_stats=True)   # S = 4*49: % 4 == 0
    y3, s3 = ops.conv3d(x[:, :, :1, :7, :7].contiguous(), wt, None, 1, (0, 1, 1), want_stats=True)  # S = 49: no
    assert s3 is None


def test_fused_multi_tensor_adam_matches_torch_adam():
    from zeroshotvideoclassification_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(6)
    shapes = [(5000,), (3, 7, 11), (1,), (4097,), (64, 64)]
    ref_p = [torch.randn(*s, generator=g).requires_grad_() for s in shapes] + [torch.randn(9).requires_grad_()]
    dev_p = [p.detach().clone().to(DEV).requires_grad_() for p in ref_p]
    ref_opt = torch.optim.Adam(ref_p, lr=1e-3)
    dev_opt = FusedAdam(dev_p, lr=1e-3)
    for step in range(3):
        for rp, dp in zip(ref_p[:-1], dev_p[:-1]):             # the last parameter never gets a gradient
            gr = torch.randn(rp.shape, generator=g)
            rp.grad = gr.clone()
            dp.grad = gr.to(DEV)
        ref_opt.step()
        dev_opt.step()
    for rp, dp in zip(ref_p, dev_p):
        close(dp, rp, rtol=1e-6, what="fused adam")
    assert torch.equal(dev_p[-1].cpu(), ref_p[-1].detach())
    sd = dev_opt.state_dict()["state"]
    assert set(sd[0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sd[0]["step"]) == 3.0


def _rand_geometry(rng):
    k = tuple(int(v) for v in rng.choice([1, 3, 5, 7], size=3, p=[0.35, 0.45, 0.1, 0.1]))
    s = tuple(int(v) for v in rng.choice([1, 2, 3], size=3, p=[0.6, 0.3, 0.1]))
    p = tuple(int(rng.integers(0, kk // 2 + 1)) for kk in k)
    dims = tuple(int(rng.integers(max(1, kk - 2 * pp), max(2, kk - 2 * pp) + 9)) for kk, pp in zip(k, p))
    cin = int(rng.choice([1, 3, 5, 16, 17, 32, 48, 70]))
    cout = int(rng.choice([1, 2, 16, 45, 64, 80, 150]))
    n = int(rng.integers(1, 4))
    return n, cin, cout, k, s, p, dims


def test_conv3d_random_geometries():
    """Fuzz over kernel / stride / padding / channel counts (incl. 1-channel, non-multiples of 16,
    voxel counts that are not multiples of 4, strides > kernel): forward, dgrad, wgrad vs fp64."""
    rng = np.random.default_rng(2024)
    g = torch.Generator().manual_seed(99)
    done = 0
    while done < 40:
        n, cin, cout, k, s, p, dims = _rand_geometry(rng)
        if any((d + 2 * pp - kk) < 0 for d, pp, kk in zip(dims, p, k)):
            continue
        x = torch.randn(n, cin, *dims, generator=g)
        wt = torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])
        use_bias = bool(rng.integers(0, 2))
        b = torch.randn(cout, generator=g) if use_bias else None
        xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
        br = b.double().requires_grad_() if use_bias else None
        yr = F.conv3d(xr, wr, br, stride=s, padding=p)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy.double())
        xg, wg = x.to(DEV).requires_grad_(), wt.to(DEV).requires_grad_()
        bg = b.to(DEV).requires_grad_() if use_bias else None
        tag = f"n{n} c{cin}->{cout} k{k} s{s} p{p} in{dims}"
        yg = ops.conv3d(xg, wg, bg, s, p)
        close(yg, yr, what=tag + " fwd")
        yg.backward(dy.to(DEV))
        close(xg.grad, xr.grad, what=tag + " dgrad")
        close(wg.grad, wr.grad, rtol=5e-5, what=tag + " wgrad")
        if use_bias:
            close(bg.grad, br.grad, what=tag + " dbias")
        done += 1


def test_batchnorm_and_pool_edge_shapes():
    g = torch.Generator().manual_seed(17)
    for shape in [(1, 1, 1, 1, 2), (2, 3, 1, 1, 1), (1, 260, 1, 3, 3), (5, 2, 3, 5, 7)]:
        x = torch.randn(*shape, generator=g)
        ref = F.batch_norm(x.double(), None, None, None, None, training=True)
        out = ops.batch_norm_act(x.to(DEV), None, None, None, None, None, True, 0.1, 1e-5, False)
        close(out, ref, rtol=1e-4, what=f"bn {shape}")
        close(ops.mean_pool(x.to(DEV)), x.double().mean(dim=(2, 3, 4)), what=f"meanpool {shape}")


WGRAD_DMA_CASES = [
    # stride 1, equal extents, T*H*W % 16 == 0, Cin >= 16: the LDS-DMA weight-gradient kernel
    # name, N, Cin, (T,H,W), Cout, kernel, padding
    ("s1_like", 2, 64, (2, 12, 12), 144, (1, 3, 3), (0, 1, 1)),           # row tile 144
    ("t1_like", 3, 144, (4, 8, 8), 64, (3, 1, 1), (1, 0, 0)),             # 3 clips: clip wrap inside a slice
    ("cin_45", 2, 45, (4, 4, 4), 64, (3, 1, 1), (1, 0, 0)),               # channel padding 45 -> 48
    ("cout_230", 1, 32, (2, 8, 8), 230, (1, 3, 3), (0, 1, 1)),            # ragged row tiles
    ("full_333", 2, 16, (4, 4, 4), 40, (3, 3, 3), (1, 1, 1)),             # 27 taps
    ("one_chunk", 1, 16, (1, 4, 4), 16, (1, 3, 3), (0, 1, 1)),            # a single 16-voxel chunk
    ("wide_w", 1, 24, (1, 4, 20), 20, (1, 3, 3), (0, 1, 1)),              # W not a multiple of 4: unaligned row shifts
    ("pointwise", 2, 64, (2, 4, 4), 128, (1, 1, 1), (0, 0, 0)),           # 1 tap, no border
    ("many_slices", 6, 32, (8, 16, 16), 48, (1, 3, 3), (0, 1, 1)),
]


@pytest.mark.parametrize("case", WGRAD_DMA_CASES, ids=[c[0] for c in WGRAD_DMA_CASES])
def test_conv3d_wgrad_dma_path(case, monkeypatch):
    name, n, cin, (t, h, w), cout, k, p = case
    g = torch.Generator().manual_seed(zlib.crc32(str(name).encode()))    # (not hash(): randomised per process, a failure could not be replayed)
    x = torch.randn(n, cin, t, h, w, generator=g)
    wt = torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * np.prod(k))
    xr, wr = x.double(), wt.double().requires_grad_()
    yr = F.conv3d(xr, wr, padding=p)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    def run():
        wg = wt.to(DEV).requires_grad_()
        ops.conv3d(x.to(DEV), wg, None, 1, p).backward(dy.to(DEV))
        return wg.grad

    a = run()
    close(a, wr.grad, what=f"{name} wgrad (dma)")
    assert torch.equal(a, run()), "bitwise reproducible"
    monkeypatch.setenv("ZSV_NO_WGRAD_DMA", "1")                     # the register-staged kernel on the same problem
    close(run(), wr.grad, what=f"{name} wgrad (register-staged)")


def test_conv3d_wgrad_dma_random_geometries():
    """Seeded sweep over the LDS-DMA weight-gradient kernel's domain (stride 1, equal extents, voxel
    count a multiple of 16, >= 256 columns): channel / row-tile padding, clip wraps, 1..27 taps, every
    column-tile width, the frame-minor chunk walk."""
    rng = np.random.RandomState(2024)
    kernels = [((1, 3, 3), (0, 1, 1)), ((3, 1, 1), (1, 0, 0)), ((3, 3, 3), (1, 1, 1)), ((1, 1, 1), (0, 0, 0)),
               ((3, 3, 1), (1, 1, 0))]
    done = 0
    while done < 24:
        k, p = kernels[rng.randint(len(kernels))]
        cin = int(rng.choice([16, 24, 45, 64, 72, 100]))
        cout = int(rng.choice([16, 40, 64, 100, 144, 230]))
        t, h, w = int(rng.choice([1, 2, 4, 8])), int(rng.choice([4, 6, 8, 12])), int(rng.choice([4, 8, 10, 16]))
        if (t * h * w) % 16 or np.prod(k) * ((cin + 15) // 16 * 16) < 256:
            continue
        n = int(rng.randint(1, 4))
        g = torch.Generator().manual_seed(1000 + done)
        x = torch.randn(n, cin, t, h, w, generator=g)
        wt = torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * np.prod(k))
        wr = wt.double().requires_grad_()
        yr = F.conv3d(x.double(), wr, padding=p)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy.double())
        wg = wt.to(DEV).requires_grad_()
        ops.conv3d(x.to(DEV), wg, None, 1, p).backward(dy.to(DEV))
        close(wg.grad, wr.grad, rtol=5e-5, what=f"wgrad dma n={n} cin={cin} cout={cout} thw={(t, h, w)} k={k}")
        done += 1


@pytest.mark.parametrize("n,cin,cout,thw", [(4, 32, 16, (8, 96, 96)), (2, 100, 24, (4, 96, 180)), (6, 64, 48, (16, 56, 56)),
                                            (4, 32, 24, (8, 64, 90)), (4, 64, 20, (8, 64, 90))],
                         ids=["one_row_tile", "two_row_tiles_padded_channels", "s1_like", "w_not_mult_of_4_48_rows",
                              "w_not_mult_of_4_64_rows"])
def test_conv3d_dgrad_winograd_path(n, cin, cout, thw, monkeypatch):
    """Input gradient of the 1x3x3 stride-1 convolutions through the Winograd F(2,3)-along-W kernel (large
    voxel counts only: it has no split-K form) against torch CPU fp64 and against the direct kernel; with
    and without the fused shortcut-gradient add."""
    import ctypes
    from zeroshotvideoclassification_amd import _lib
    t, h, w = thw
    g = torch.Generator().manual_seed(cin * 7 + cout)
    wt = torch.randn(cout, cin, 1, 3, 3, generator=g) / np.sqrt(cin * 9)
    dy = torch.randn(n, cout, t, h, w, generator=g)
    add = torch.randn(n, cin, t, h, w, generator=g)
    ref = torch.nn.grad.conv3d_input((n, cin, t, h, w), wt.double(), dy.double(), stride=1, padding=(0, 1, 1))
    lib = _lib.load()
    d = ops.conv_desc((n, cin, t, h, w), wt.shape, 1, (0, 1, 1))

    def run(with_add):
        dx = torch.empty((n, cin, t, h, w), device=DEV)
        nbytes = lib.zsv_conv3d_dgrad_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=DEV)
        a = add.to(DEV) if with_add else None
        _lib.check(lib.zsv_conv3d_dgrad_add(ctypes.byref(d), dy.to(DEV).data_ptr(), wt.to(DEV).data_ptr(),
                                            a.data_ptr() if a is not None else None, dx.data_ptr(), ws.data_ptr(), nbytes,
                                            None), "dgrad")
        torch.cuda.synchronize()
        return dx

    close(run(False), ref, what="winograd dgrad")
    close(run(True), ref + add.double(), what="winograd dgrad + add")
    fast = run(False)
    monkeypatch.setenv("ZSV_NO_WINO", "1")
    direct = run(False)
    close(fast, direct.double(), rtol=wino_vs_direct_tol(w), what="winograd vs direct kernel")
    assert not torch.equal(fast, direct), "the two paths should not be the same kernel"


@pytest.mark.parametrize("n,cin,cout,thw", [(4, 16, 32, (8, 96, 96)), (2, 24, 100, (4, 96, 180)), (6, 64, 144, (16, 56, 56)),
                                            (4, 45, 288, (8, 28, 56)), (4, 20, 40, (8, 64, 90)), (4, 24, 64, (8, 64, 90)),
                                            (22, 256, 460, (4, 14, 14))],
                         ids=["one_48_row_tile", "padded_channels_64_rows", "s1_48_row_tiles", "288_rows_ragged_k",
                              "w_not_mult_of_4_48_rows", "w_not_mult_of_4_64_rows", "layer3_shape"])
def test_conv3d_fwd_winograd_path(n, cin, cout, thw, monkeypatch):
    """Forward of the 1x3x3 stride-1 convolutions through the Winograd F(2,3)-along-W kernel (64- and 48-row
    tiles) against torch CPU fp64 and the direct kernel: plain, with the BatchNorm partial statistics of the
    training forward, and with the inference epilogue (bias + residual + ReLU)."""
    import ctypes
    from zeroshotvideoclassification_amd import _lib
    t, h, w = thw
    g = torch.Generator().manual_seed(cin * 11 + cout)
    wt = torch.randn(cout, cin, 1, 3, 3, generator=g) / np.sqrt(cin * 9)
    x = torch.randn(n, cin, t, h, w, generator=g)
    bias = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, t, h, w, generator=g)
    ref = F.conv3d(x.double(), wt.double(), padding=(0, 1, 1))
    lib = _lib.load()
    d = ops.conv_desc(x.shape, wt.shape, 1, (0, 1, 1))
    xd, wd, bd, rd = x.to(DEV), wt.to(DEV), bias.to(DEV), res.to(DEV)

    def run(mode):
        y = torch.empty((n, cout, t, h, w), device=DEV)
        nbytes = lib.zsv_conv3d_fwd_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=DEV)
        stats, tiles = None, 0
        if mode == "stats":
            tiles = lib.zsv_conv3d_fwd_stat_tiles(ctypes.byref(d), y.data_ptr())
            assert tiles > 0
            stats = torch.full((2, cout, tiles), float("nan"), device=DEV)
        full = mode == "full"
        _lib.check(lib.zsv_conv3d_fwd_full(ctypes.byref(d), xd.data_ptr(), wd.data_ptr(), bd.data_ptr() if full else None,
                                           rd.data_ptr() if full else None, y.data_ptr(), 1 if full else 0,
                                           stats.data_ptr() if stats is not None else None, tiles, ws.data_ptr(), nbytes,
                                           None), "fwd")
        torch.cuda.synchronize()
        return y, stats

    fast, _ = run("plain")
    close(fast, ref, what="winograd fwd")
    y, stats = run("stats")
    assert torch.equal(y, fast)
    close(stats[0].double().sum(1), ref.sum(dim=(0, 2, 3, 4)), rtol=1e-4, what="winograd fwd: sum y")
    close(stats[1].double().sum(1), (ref * ref).sum(dim=(0, 2, 3, 4)), rtol=1e-4, what="winograd fwd: sum y^2")
    y, _ = run("full")
    close(y, torch.relu(ref + bias.double().view(1, -1, 1, 1, 1) + res.double()), what="winograd fwd + bias + residual + relu")
    if w % 4 == 0:                                             # F(4,3) (shipped for W % 4 == 0) against the F(2,3) kernel
        monkeypatch.setenv("ZSV_WINO_NO_F43", "1")
        f23, _ = run("plain")
        monkeypatch.delenv("ZSV_WINO_NO_F43")
        close(fast, f23.double(), rtol=1e-5, what="F(4,3) vs F(2,3)")
        assert not torch.equal(fast, f23), "F(4,3) and F(2,3) should not be the same kernel"
    monkeypatch.setenv("ZSV_NO_WINO_FWD", "1")
    direct, _ = run("plain")
    close(fast, direct.double(), rtol=wino_vs_direct_tol(w), what="winograd vs direct kernel")
    assert not torch.equal(fast, direct), "the two paths should not be the same kernel"


@pytest.mark.parametrize("n,cin,cout,thw", [(4, 16, 32, (8, 64, 96)), (6, 24, 64, (5, 64, 90)), (2, 64, 128, (16, 56, 56))],
                         ids=["small_channels", "odd_frames_w_not_mult_of_4", "c3d_conv2_like"])
def test_conv3d_333_winograd_path(n, cin, cout, thw, monkeypatch):
    """3x3x3 stride-1 "same" convolutions (C3D network.py:102-117, Conv3DSimple resnet.py:23-30) through the
    Winograd kernel (9 (kt, kh) row taps): relu(conv + bias) forward and both gradients through autograd
    against torch CPU fp64, and forward / input gradient against the direct kernel."""
    t, h, w = thw
    g = torch.Generator().manual_seed(cin * 13 + cout)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) / np.sqrt(cin * 27)
    x = torch.randn(n, cin, t, h, w, generator=g)
    bias = torch.randn(cout, generator=g) * 0.1
    dy = torch.randn(n, cout, t, h, w, generator=g)
    yr = torch.relu(F.conv3d(x.double(), wt.double(), bias.double(), padding=1))

    def run():
        xg, wg, bg = x.to(DEV).requires_grad_(), wt.to(DEV).requires_grad_(), bias.to(DEV).requires_grad_()
        y = ops.conv3d(xg, wg, bg, 1, 1, relu=True)
        y.backward(dy.to(DEV))
        return y.detach(), xg.grad, wg.grad, bg.grad

    y, dx, dw, db = run()
    close(y, yr, what="3x3x3 winograd fwd (+bias, relu)")
    # the gradients' reference uses the device's own ReLU mask (outputs within rounding of 0 may differ in sign)
    gm = dy.double() * (y.cpu() > 0)
    close(dx, torch.nn.grad.conv3d_input(x.shape, wt.double(), gm, padding=1), what="3x3x3 winograd dgrad")
    close(dw, torch.nn.grad.conv3d_weight(x.double(), wt.shape, gm, padding=1), rtol=5e-5, what="3x3x3 wgrad")
    close(db, gm.sum(dim=(0, 2, 3, 4)), rtol=5e-5, what="3x3x3 dbias")
    monkeypatch.setenv("ZSV_NO_WINO", "1")
    y2, dx2, _, _ = run()
    close(y, y2.double(), rtol=wino_vs_direct_tol(w), what="winograd vs direct kernel (fwd)")
    if torch.equal(y > 0, y2 > 0):
        close(dx, dx2.double(), rtol=wino_vs_direct_tol(w), what="winograd vs direct kernel (dgrad)")
    assert not torch.equal(y, y2), "the two paths should not be the same kernel"


WGRAD_WINO_CASES = [
    # name, N, Cin, (T,H,W), Cout, kT
    ("s1_like", 3, 64, (4, 56, 56), 144, 1),                 # 144-row tile, 3 column tiles
    ("m230_cin45", 6, 45, (8, 28, 28), 230, 1),              # two 128-row tiles (ragged), channel padding, ragged column tile
    ("c3d_333", 2, 32, (8, 32, 64), 128, 3),                 # 9 row taps
    ("two_frames_333", 8, 16, (2, 32, 64), 120, 3),          # every voxel touches a temporal border
    ("layer3_14x14", 22, 32, (4, 14, 14), 144, 1),           # W % 4 = 2 (unaligned X pieces), S = 784: partial last chunk
    ("w_10_odd_chunks", 30, 16, (3, 20, 10), 128, 3),        # S = 600 = 18.75 chunks
    ("r3d_layer1_like_64_rows", 2, 64, (8, 32, 64), 64, 3),    # 64-row tile, two column blocks per wave
    ("m56_ragged_columns", 4, 24, (4, 32, 40), 56, 1),          # 64-row tile, 96 of 128 columns, ragged rows
    ("m64_14x14", 22, 32, (4, 14, 14), 64, 1),                  # 64-row tile with the edge handling
    ("partial_chunk_aligned_w", 24, 16, (3, 20, 12), 128, 1),  # S = 720 = 22.5 chunks, W % 4 = 0
    ("unaligned_w_whole_chunks", 28, 16, (4, 16, 10), 128, 1), # S = 640 = 20 chunks, W % 4 = 2
    ("two_144_row_tiles", 6, 32, (4, 28, 28), 288, 1),         # F(4,3) form: 5 + 4 row blocks per workgroup, two row tiles
    ("m200_ragged_second_tile", 12, 48, (2, 24, 32), 200, 3),   # F(4,3) form: 144 + 56 rows, 9 row taps
]


@pytest.mark.parametrize("case", WGRAD_WINO_CASES, ids=[c[0] for c in WGRAD_WINO_CASES])
def test_conv3d_wgrad_winograd_path(case, monkeypatch):
    """Weight gradient of the 1x3x3 / 3x3x3 stride-1 convolutions through the Winograd-form kernel
    (conv_wgrad_wino.hip) against torch CPU fp64 and the plain LDS-DMA kernel; bitwise reproducible."""
    name, n, cin, (t, h, w), cout, kt = case
    g = torch.Generator().manual_seed(len(name) * 101 + cin)
    x = torch.randn(n, cin, t, h, w, generator=g)
    dy = torch.randn(n, cout, t, h, w, generator=g)
    wshape = (cout, cin, kt, 3, 3)
    pad = (kt // 2, 1, 1)
    ref = torch.nn.grad.conv3d_weight(x.double(), wshape, dy.double(), padding=pad)

    def run():
        wg = torch.zeros(wshape, device=DEV).requires_grad_()
        ops.conv3d(x.to(DEV), wg, None, 1, pad).backward(dy.to(DEV))
        return wg.grad

    a = run()
    close(a, ref, what=f"{name} wgrad (winograd)")
    assert torch.equal(a, run()), "bitwise reproducible"
    if w % 4 == 0 and (t * h * w) % 32 == 0 and cout > 64:      # the F(4,3) form (conv_wgrad_wino4_kernel) took it: also the F(2,3) form
        monkeypatch.setenv("ZSV_NO_WGRAD_WINO4", "1")
        a2 = run()
        close(a2, ref, what=f"{name} wgrad (winograd, F(2,3) form)")
        assert not torch.equal(a, a2), "F(4,3) and F(2,3) forms should not be the same kernel"
    monkeypatch.setenv("ZSV_NO_WGRAD_WINO", "1")
    b = run()
    close(b, ref, what=f"{name} wgrad (plain)")
    assert not torch.equal(a, b), "the two paths should not be the same kernel"


FULL_SIZE_LAYERS = [
    # BASELINE.json configs[1] / configs[3] geometries at the benchmark's 22 clips: name, Cin, Cout, kernel, padding, (T,H,W)
    ("S1_spatial_64_144", 64, 144, (1, 3, 3), (0, 1, 1), (16, 56, 56)),
    ("T1_temporal_144_64", 144, 64, (3, 1, 1), (1, 0, 0), (16, 56, 56)),
    ("S4_spatial_128_288", 128, 288, (1, 3, 3), (0, 1, 1), (8, 28, 28)),
    ("S7_spatial_256_576", 256, 576, (1, 3, 3), (0, 1, 1), (4, 14, 14)),
    ("C2_c3d_64_128", 64, 128, (3, 3, 3), (1, 1, 1), (16, 56, 56)),
]


@pytest.mark.parametrize("case", FULL_SIZE_LAYERS, ids=[c[0] for c in FULL_SIZE_LAYERS])
def test_full_size_adjoint_identities(case):
    """Size-independent property at the benchmark's full size (N = 22), where a CPU fp64 convolution is
    too slow to be the checker: forward, input gradient and weight gradient are the three faces of one
    trilinear form, so <conv(x, w), g> = <x, dgrad(g, w)> = <w, wgrad(x, g)> -- checked with fp64 dot
    products; plus the forward against torch CPU fp64 on one clip of the batch."""
    name, cin, cout, k, p, (t, h, w) = case
    n = 22
    g = torch.Generator(device=DEV).manual_seed(len(name))
    x = torch.randn(n, cin, t, h, w, device=DEV, generator=g).requires_grad_()
    wt = (torch.randn(cout, cin, *k, device=DEV, generator=g) / np.sqrt(cin * np.prod(k))).requires_grad_()
    gy = torch.randn(n, cout, t, h, w, device=DEV, generator=g)
    y = ops.conv3d(x, wt, None, 1, p)
    y.backward(gy)
    a = (y.detach().double() * gy.double()).sum().item()
    b = (x.detach().double() * x.grad.double()).sum().item()
    c = (wt.detach().double() * wt.grad.double()).sum().item()
    scale = (y.detach().double().norm() * gy.double().norm()).item()
    assert abs(a - b) <= 2e-6 * scale, f"{name}: <y,g>={a!r} vs <x,dx>={b!r} (scale {scale:.3e})"
    assert abs(a - c) <= 2e-6 * scale, f"{name}: <y,g>={a!r} vs <w,dw>={c!r} (scale {scale:.3e})"
    i = n - 1                                               # last clip: the tail tiles
    ref = F.conv3d(x.detach()[i:i + 1].cpu().double(), wt.detach().cpu().double(), padding=p)
    close(y.detach()[i:i + 1], ref, what=f"{name} forward, clip {i}")


@pytest.mark.parametrize("cin,cout,k,use_bias,hw", [(256, 256, (1, 3, 3), False, 14), (256, 256, (3, 3, 3), True, 14),
                                                    (460, 256, (1, 3, 3), False, 14), (256, 256, (1, 3, 3), True, 16)],
                         ids=["layer3_1x3x3", "r3d_layer3_3x3x3_bias_relu", "ragged_k_460", "w16_f43_form"])
def test_conv3d_winograd_split_k(cin, cout, k, use_bias, hw, monkeypatch):
    """Layer3-sized problems (22 clips of 4x14x14: 272 tiles, fewer than one round of workgroups) run the Winograd
    kernel in K parts whose partial results are summed in order: forward (bias + ReLU in the sum) and input
    gradient against torch CPU fp64 on the first and last clip and against the direct kernel on all of them."""
    n, t, h, w = 22, 4, hw, hw
    p = (k[0] // 2, 1, 1)
    g = torch.Generator().manual_seed(cin + cout + k[0])
    x = torch.randn(n, cin, t, h, w, generator=g)
    wt = torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * np.prod(k))
    bias = torch.randn(cout, generator=g) * 0.1 if use_bias else None
    dy = torch.randn(n, cout, t, h, w, generator=g)

    def run():
        xg = x.to(DEV).requires_grad_()
        y = ops.conv3d(xg, wt.to(DEV), bias.to(DEV) if use_bias else None, 1, p, relu=use_bias)
        y.backward(dy.to(DEV))
        return y.detach(), xg.grad

    y, dx = run()
    assert torch.equal(y, run()[0]), "bitwise reproducible"
    for i in (0, n - 1):
        ref = F.conv3d(x[i:i + 1].double(), wt.double(), bias.double() if use_bias else None, padding=p)
        if use_bias:
            ref = torch.relu(ref)
        close(y[i:i + 1], ref, what=f"split-K winograd forward, clip {i}")
        gm = dy[i:i + 1].double() * ((y[i:i + 1].cpu() > 0) if use_bias else 1.0)
        close(dx[i:i + 1], torch.nn.grad.conv3d_input((1, cin, t, h, w), wt.double(), gm, padding=p),
              what=f"split-K winograd dgrad, clip {i}")
    monkeypatch.setenv("ZSV_NO_WINO", "1")
    y2, dx2 = run()
    close(y, y2.double(), rtol=wino_vs_direct_tol(w), what="winograd (K parts) vs direct kernel, forward")
    if torch.equal(y > 0, y2 > 0):
        close(dx, dx2.double(), rtol=wino_vs_direct_tol(w), what="winograd (K parts) vs direct kernel, dgrad")
    assert not torch.equal(dx, dx2), "the two paths should not be the same kernel"


WGRAD_TRING_CASES = [
    # name, N, Cin, (T,H,W), Cout
    ("t1_like", 2, 144, (8, 32, 32), 64),                    # one 144-row tile, one column tile
    ("cin230_cout128", 4, 230, (4, 32, 32), 128),            # two 128-row tiles (ragged), two column tiles, T = 4
    ("ragged_both", 2, 130, (16, 16, 64), 56),               # channel padding on both sides
    ("many_clips", 12, 144, (4, 16, 28), 64),                # HW = 448 = 28 segments; slices start mid-segment
]


@pytest.mark.parametrize("case", WGRAD_TRING_CASES, ids=[c[0] for c in WGRAD_TRING_CASES])
def test_conv3d_wgrad_temporal_ring_path(case, monkeypatch):
    """Weight gradient of the temporal 3x1x1 stride-1 convolutions through the dY-frame-ring kernel
    (conv_wgrad_tring.hip) against torch CPU fp64 and the plain LDS-DMA kernel; bitwise reproducible."""
    name, n, cin, (t, h, w), cout = case
    g = torch.Generator().manual_seed(len(name) * 17 + cin)
    x = torch.randn(n, cin, t, h, w, generator=g)
    dy = torch.randn(n, cout, t, h, w, generator=g)
    wshape = (cout, cin, 3, 1, 1)
    ref = torch.nn.grad.conv3d_weight(x.double(), wshape, dy.double(), padding=(1, 0, 0))

    def run():
        wg = torch.zeros(wshape, device=DEV).requires_grad_()
        ops.conv3d(x.to(DEV), wg, None, 1, (1, 0, 0)).backward(dy.to(DEV))
        return wg.grad

    a = run()
    close(a, ref, what=f"{name} wgrad (frame ring)")
    assert torch.equal(a, run()), "bitwise reproducible"
    monkeypatch.setenv("ZSV_NO_WGRAD_TRING", "1")
    b = run()
    close(b, ref, what=f"{name} wgrad (plain)")
    # (both kernels add the voxels of a slice in the same order, so with equal slice counts they can agree bit for bit)
    monkeypatch.delenv("ZSV_NO_WGRAD_TRING")
    monkeypatch.setenv("ZSV_WGRAD_TRING_SLICES", "7")
    c = run()
    close(c, ref, what=f"{name} wgrad (frame ring, 7 slices)")
    assert not torch.equal(a, c), "the slice-count knob of the ring kernel had no effect: is the kernel in use?"


@pytest.mark.parametrize("n,cin,cout,thw", [(2, 144, 64, (8, 32, 32)), (3, 230, 128, (4, 32, 48)), (2, 130, 56, (16, 16, 64))],
                         ids=["t1_like", "ragged_230_to_128", "padded_rows_and_columns"])
def test_batchnorm_relu_folded_into_the_temporal_convolution(n, cin, cout, thw, monkeypatch):
    """BatchNorm3d -> ReLU -> Conv3d(3x1x1) (Conv2Plus1D's mid tensor, resnet.py:46-52) with the normalise pass folded into
    the convolution's forward and weight-gradient kernels: bit-identical to the unfused sequence (outputs, every gradient,
    running statistics), and equal to the fp64 reference."""
    from zeroshotvideoclassification_amd import layers
    t, h, w = thw
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, t, h, w, generator=g) * 1.5 + 0.3
    wt = torch.randn(cout, cin, 3, 1, 1, generator=g) / np.sqrt(cin * 3)
    gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.2
    dy = torch.randn(n, cout, t, h, w, generator=g)
    assert ops.conv_pre_supported(x.shape, wt.shape, 1, (1, 0, 0))

    def run(fused):
        bn = layers.BatchNorm3d(cin).to(DEV).train()
        bn.weight.data.copy_(gamma)
        bn.bias.data.copy_(beta)
        conv = layers.Conv3d(cin, cout, kernel_size=(3, 1, 1), padding=(1, 0, 0), bias=False).to(DEV)
        conv.weight.data.copy_(wt)
        xg = x.to(DEV).requires_grad_()
        mask = None
        if fused:
            handle, coef = bn.deferred(xg)
            y, stats = conv.forward_pre(handle, coef, want_stats=True)
        else:
            act = bn(xg, relu=True)
            mask = (act.detach() > 0).cpu()
            y, stats = conv(act, want_stats=True)
        y.backward(dy.to(DEV))
        torch.cuda.synchronize()
        return dict(y=y.detach(), stats=stats, dx=xg.grad, dgamma=bn.weight.grad, dbeta=bn.bias.grad, dw=conv.weight.grad,
                    rm=bn.running_mean.clone(), rv=bn.running_var.clone(), nbt=bn.num_batches_tracked.clone()), mask

    (a, _), (b, mask) = run(True), run(False)
    for k in a:
        assert (a[k] is None) == (b[k] is None), k
        if a[k] is not None:
            assert torch.equal(a[k], b[k]), f"fused vs unfused differ in {k}"
    xr = x.double().requires_grad_()
    gr, br, wr = gamma.double().requires_grad_(), beta.double().requires_grad_(), wt.double().requires_grad_()
    # (the reference uses the device's own ReLU mask: a pre-activation within rounding of 0 may differ in sign)
    yr = F.conv3d(F.batch_norm(xr, None, None, gr, br, training=True) * mask.double(), wr, padding=(1, 0, 0))
    yr.backward(dy.double())
    close(a["y"], yr, what="folded bn+relu+conv forward")
    close(a["dx"], xr.grad, rtol=1e-4, what="folded dx")
    close(a["dw"], wr.grad, rtol=5e-5, what="folded dw")
    close(a["dgamma"], gr.grad, rtol=1e-4, what="folded dgamma")
    close(a["dbeta"], br.grad, rtol=1e-4, what="folded dbeta")
    monkeypatch.setenv("ZSV_NO_BN_FUSION", "1")
    assert not ops.conv_pre_supported(x.shape, wt.shape, 1, (1, 0, 0))
    # geometries without a fused path say so (strided, spatial taps, too few voxels) and the chain falls back
    assert not ops.conv_pre_supported((2, 64, 8, 32, 32), (144, 64, 1, 3, 3), 1, (0, 1, 1))


@pytest.mark.parametrize("n,cin,cout,thw", [(4, 144, 64, (16, 56, 56)), (22, 230, 128, (8, 28, 28)), (6, 48, 144, (16, 40, 40)),
                                            (5, 64, 45, (16, 48, 48)), (22, 128, 256, (4, 14, 14)), (3, 20, 70, (8, 60, 60)),
                                            (3, 144, 64, (32, 28, 28)), (5, 48, 144, (32, 22, 22))],
                         ids=["t1_like", "layer2_ragged_last_segment", "three_row_tiles", "ragged_rows_45", "layer3_t4_partial_tiles",
                              "ragged_k_20_channels", "t32_two_frame_quads_per_wave", "t32_ragged_last_segment_three_row_tiles"])
def test_conv3d_temporal_winograd_path(n, cin, cout, thw, monkeypatch):
    """Temporal 3x1x1 stride-1 convolutions through the Winograd-along-T kernels (conv_winot4_kernel, F(4,3), and conv_winot_kernel,
    F(2,3)): forward (plain, with the BatchNorm partial statistics) and input gradient (plain, with the fused shortcut add) against
    torch CPU fp64, against each other and against the direct kernel; the folded BatchNorm + ReLU prologue bit-identical to the
    separate pass."""
    import ctypes
    from zeroshotvideoclassification_amd import _lib, layers
    t, h, w = thw
    g = torch.Generator().manual_seed(cin * 3 + cout)
    wt = torch.randn(cout, cin, 3, 1, 1, generator=g) / np.sqrt(cin * 3)
    x = torch.randn(n, cin, t, h, w, generator=g)
    dy = torch.randn(n, cout, t, h, w, generator=g)
    add = torch.randn(n, cin, t, h, w, generator=g)
    yr = F.conv3d(x.double(), wt.double(), padding=(1, 0, 0))
    dxr = torch.nn.grad.conv3d_input(x.shape, wt.double(), dy.double(), padding=(1, 0, 0))
    lib = _lib.load()
    d = ops.conv_desc(x.shape, wt.shape, 1, (1, 0, 0))
    xd, wd, dyd, addd = x.to(DEV), wt.to(DEV), dy.to(DEV), add.to(DEV)

    def fwd(want_stats):
        y = torch.empty((n, cout, t, h, w), device=DEV)
        nbytes = lib.zsv_conv3d_fwd_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=DEV)
        stats, tiles = None, 0
        if want_stats:
            tiles = lib.zsv_conv3d_fwd_stat_tiles(ctypes.byref(d), y.data_ptr())
            assert tiles > 0
            stats = torch.full((2, cout, tiles), float("nan"), device=DEV)
        _lib.check(lib.zsv_conv3d_fwd_stats(ctypes.byref(d), xd.data_ptr(), wd.data_ptr(), None, y.data_ptr(), 0,
                                            stats.data_ptr() if stats is not None else None, tiles, ws.data_ptr(), nbytes, None), "fwd")
        torch.cuda.synchronize()
        return y, stats

    def dgrad(with_add):
        dx = torch.empty((n, cin, t, h, w), device=DEV)
        nbytes = lib.zsv_conv3d_dgrad_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=DEV)
        _lib.check(lib.zsv_conv3d_dgrad_add(ctypes.byref(d), dyd.data_ptr(), wd.data_ptr(), addd.data_ptr() if with_add else None,
                                            dx.data_ptr(), ws.data_ptr(), nbytes, None), "dgrad")
        torch.cuda.synchronize()
        return dx

    y, _ = fwd(False)
    close(y, yr, what="temporal winograd fwd")
    y2, stats = fwd(True)
    assert torch.equal(y, y2)
    close(stats[0].double().sum(1), yr.sum(dim=(0, 2, 3, 4)), rtol=1e-4, what="temporal winograd fwd: sum y")
    close(stats[1].double().sum(1), (yr * yr).sum(dim=(0, 2, 3, 4)), rtol=1e-4, what="temporal winograd fwd: sum y^2")
    dx = dgrad(False)
    close(dx, dxr, what="temporal winograd dgrad")
    if lib.zsv_conv3d_dgrad_add_supported(ctypes.byref(d)):          # (not where the dgrad runs in K parts on the direct kernel)
        close(dgrad(True), dxr + add.double(), what="temporal winograd dgrad + add")
    # folded BatchNorm + ReLU (PRE form of this kernel) vs the separate normalise pass: same bits
    if ops.conv_pre_supported(x.shape, wt.shape, 1, (1, 0, 0)):
        def chain(fused):
            bn = layers.BatchNorm3d(cin).to(DEV).train()
            conv = layers.Conv3d(cin, cout, kernel_size=(3, 1, 1), padding=(1, 0, 0), bias=False).to(DEV)
            conv.weight.data.copy_(wt)
            xg = x.to(DEV).requires_grad_()
            if fused:
                handle, coef = bn.deferred(xg)
                out = conv.forward_pre(handle, coef)
            else:
                out = conv(bn(xg, relu=True))
            out.backward(dyd)
            torch.cuda.synchronize()
            return out.detach(), xg.grad, conv.weight.grad, bn.weight.grad
        for a, b in zip(chain(True), chain(False)):
            assert torch.equal(a, b), "folded BatchNorm differs from the separate pass"
    # F(4,3) (default) against F(2,3) along T, then both against the direct kernel (F(4,3): transform constants up to 8, tested at 1e-5)
    monkeypatch.setenv("ZSV_WINOT_NO_F43", "1")
    y23, stats23 = fwd(True)
    dx23 = dgrad(False)
    close(y23, yr, what="temporal F(2,3) fwd")
    close(dx23, dxr, what="temporal F(2,3) dgrad")
    close(stats23[0].double().sum(1), yr.sum(dim=(0, 2, 3, 4)), rtol=1e-4, what="temporal F(2,3) fwd: sum y")
    close(y, y23.double(), rtol=1e-5, what="temporal F(4,3) vs F(2,3) (fwd)")
    close(dx, dx23.double(), rtol=1e-5, what="temporal F(4,3) vs F(2,3) (dgrad)")
    assert not torch.equal(y, y23), "the two forms should not be the same kernel"
    monkeypatch.delenv("ZSV_WINOT_NO_F43")
    monkeypatch.setenv("ZSV_NO_WINOT", "1")
    yd, _ = fwd(False)
    dxd = dgrad(False)
    close(y, yd.double(), rtol=1e-5, what="temporal winograd vs direct kernel (fwd)")
    close(dx, dxd.double(), rtol=1e-5, what="temporal winograd vs direct kernel (dgrad)")
    close(y23, yd.double(), rtol=5e-6, what="temporal F(2,3) vs direct kernel (fwd)")
    close(dx23, dxd.double(), rtol=5e-6, what="temporal F(2,3) vs direct kernel (dgrad)")
    assert not torch.equal(y, yd), "the two paths should not be the same kernel (forward)"


S2_DGRAD_CASES = [
    # name, N, Cin, Cout, (T, H, W) of x, kind
    ("s2_like", 3, 64, 230, (8, 56, 56), "hw"),
    ("s5_like", 4, 128, 460, (8, 28, 28), "hw"),
    ("s8_full_size_k_parts", 22, 256, 921, (4, 14, 14), "hw"),
    ("ragged_rows_and_k", 3, 70, 37, (3, 10, 12), "hw"),
    ("ragged_16_byte_pieces", 3, 70, 37, (2, 12, 16), "hw"),
    ("long_rows", 2, 16, 24, (2, 6, 254), "hw"),
    ("one_clip_one_frame", 1, 16, 8, (1, 2, 2), "hw"),
    ("t2_like", 3, 230, 128, (16, 28, 28), "t"),
    ("t8_full_size_k_parts", 22, 921, 512, (4, 7, 7), "t"),
    ("temporal_ragged", 3, 50, 19, (6, 5, 7), "t"),
    ("temporal_ragged_16_byte_pieces", 3, 50, 19, (4, 6, 6), "t"),
    ("temporal_two_frames", 2, 16, 16, (2, 3, 3), "t"),
]


@pytest.mark.parametrize("case", S2_DGRAD_CASES, ids=[c[0] for c in S2_DGRAD_CASES])
def test_conv3d_stride2_dgrad_all_classes_in_one_launch(case, monkeypatch):
    """Input gradient of Conv2Plus1D's strided convolutions (1x3x3 stride (1,2,2), 3x1x1 stride (2,1,1); resnet.py:40-52 with
    the strides of :217-220) through conv_dgrad_s2.hip -- every residue class of input voxels in one launch -- against torch
    CPU fp64 and against the class-by-class launches of the direct kernel, for both tile widths and with the K range cut
    into parts."""
    import ctypes
    from zeroshotvideoclassification_amd import _lib
    name, n, cin, cout, (t, h, w), kind = case
    k, s, p = ((1, 3, 3), (1, 2, 2), (0, 1, 1)) if kind == "hw" else ((3, 1, 1), (2, 1, 1), (1, 0, 0))
    g = torch.Generator().manual_seed(cin * 11 + cout)
    wt = torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])
    d = ops.conv_desc((n, cin, t, h, w), wt.shape, s, p)
    dy = torch.randn(n, cout, d.To, d.Ho, d.Wo, generator=g)
    ref = torch.nn.grad.conv3d_input((n, cin, t, h, w), wt.double(), dy.double(), stride=s, padding=p)
    lib = _lib.load()
    dy_d, wt_d = dy.to(DEV), wt.to(DEV)

    def run():
        dx = torch.full((n, cin, t, h, w), float("nan"), device=DEV)          # every voxel must be written
        nbytes = lib.zsv_conv3d_dgrad_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=DEV)
        _lib.check(lib.zsv_conv3d_dgrad(ctypes.byref(d), dy_d.data_ptr(), wt_d.data_ptr(), dx.data_ptr(), ws.data_ptr(), nbytes,
                                        None), "dgrad")
        torch.cuda.synchronize()
        return dx

    merged = run()
    close(merged, ref, what=f"{name}: merged-class dgrad")
    assert torch.equal(merged, run()), "bitwise reproducible"
    for bn in ("64", "128"):
        for ks in ("1", "2", "3"):
            monkeypatch.setenv("ZSV_DGRAD_S2_BN", bn)
            monkeypatch.setenv("ZSV_DGRAD_S2_KS", ks)
            close(run(), ref, what=f"{name}: tile width {bn}, {ks} K parts")
    monkeypatch.delenv("ZSV_DGRAD_S2_BN")
    monkeypatch.delenv("ZSV_DGRAD_S2_KS")
    monkeypatch.setenv("ZSV_DGRAD_S2_NO_X4", "1")             # 4-byte image DMAs (what clips of S % 4 != 0 voxels use)
    assert torch.equal(run(), merged), "the DMA width does not change the arithmetic"
    monkeypatch.delenv("ZSV_DGRAD_S2_NO_X4")
    monkeypatch.setenv("ZSV_NO_DGRAD_S2", "1")
    per_class = run()
    close(merged, per_class.double(), rtol=5e-6, what=f"{name}: merged vs class-by-class launches")
    if cout >= 32:                                          # (short sums can round identically)
        assert not torch.equal(merged, per_class), "the two paths should not be the same kernel"


@pytest.mark.parametrize("n,t,h,w,cout", [(2, 4, 56, 56, 45), (3, 2, 112, 112, 45), (1, 1, 8, 8, 45), (2, 3, 24, 40, 20), (2, 2, 30, 128, 48)],
                         ids=["half_size", "full_frames", "one_tiny_frame", "fewer_output_channels", "longest_rows_odd_row_count"])
def test_stem_weight_gradient_row_kernel(n, t, h, w, cout, monkeypatch):
    """Weight gradient of the R(2+1)D stem (resnet.py:170: 3 -> 45, (1,7,7), stride (1,2,2), padding (0,3,3)) through
    stem_wgrad_kernel (operands as rows, the seven input rows of an output row in a ring) against torch CPU fp64 and against
    the generic per-element kernel."""
    g = torch.Generator().manual_seed(n * 100 + h)
    x = torch.randn(n, 3, t, h, w, generator=g)
    wt = torch.randn(cout, 3, 1, 7, 7, generator=g) * 0.1
    xr, wr = x.double(), wt.double().requires_grad_()
    yr = F.conv3d(xr, wr, stride=(1, 2, 2), padding=(0, 3, 3))
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    def run():
        xg, wg = x.to(DEV), wt.to(DEV).requires_grad_()
        ops.conv3d(xg, wg, None, (1, 2, 2), (0, 3, 3)).backward(dy.to(DEV))
        torch.cuda.synchronize()
        return wg.grad

    rows = run()
    close(rows, wr.grad, rtol=5e-5, what="stem wgrad (row kernel)")
    assert torch.equal(rows, run()), "bitwise reproducible"
    monkeypatch.setenv("ZSV_NO_STEM_WGRAD", "1")
    generic = run()
    close(rows, generic.double(), rtol=2e-5, what="row kernel vs generic kernel")
    if n * t * h * w >= 4096:                                  # (sums of a few terms can round identically)
        assert not torch.equal(rows, generic), "the two paths should not be the same kernel"


def test_conv3d_r2plus1d_family_fuzz():
    """Random extents / channel counts of the geometries that have kernels of their own -- the strided 1x3x3 and 3x1x1 convolutions
    (all residue classes in one launch), the stride-1 temporal convolution on whole frame quads (Winograd along T, incl. ragged
    position segments), the 1x3x3 stride-1 convolution (Winograd along W) and the 3-channel 7x7 stem (row kernel for its wgrad):
    forward, input gradient and weight gradient through ops.conv3d against torch CPU fp64."""
    rng = np.random.default_rng(77)
    g = torch.Generator().manual_seed(123)
    chans = [16, 24, 45, 64, 70, 128, 144, 230]
    cases = []
    for _ in range(6):      # strided spatial
        cases.append(((1, 3, 3), (1, 2, 2), (0, 1, 1), int(rng.integers(1, 4)), int(rng.choice(chans)), int(rng.choice(chans)),
                      (int(rng.integers(1, 6)), 2 * int(rng.integers(1, 21)), 2 * int(rng.integers(1, 21)))))
    for _ in range(5):      # strided temporal
        cases.append(((3, 1, 1), (2, 1, 1), (1, 0, 0), int(rng.integers(1, 4)), int(rng.choice(chans)), int(rng.choice(chans)),
                      (2 * int(rng.integers(1, 7)), int(rng.integers(1, 15)), int(rng.integers(1, 15)))))
    for _ in range(6):      # stride-1 temporal, T in {4, 8, 16}, H*W % 4 == 0, enough tiles for the Winograd path
        t = int(rng.choice([4, 8, 16]))
        cases.append(((3, 1, 1), (1, 1, 1), (1, 0, 0), int(rng.integers(3, 7)), int(rng.choice(chans)), int(rng.choice(chans)),
                      (t, 2 * int(rng.integers(10, 25)), 2 * int(rng.integers(10, 25)))))
    for _ in range(4):      # stride-1 spatial, even W (F(2,3)) or W % 4 == 0 (F(4,3))
        cases.append(((1, 3, 3), (1, 1, 1), (0, 1, 1), int(rng.integers(2, 5)), int(rng.choice(chans)), int(rng.choice(chans)),
                      (int(rng.integers(2, 7)), int(rng.integers(20, 50)), 2 * int(rng.integers(12, 40)))))
    for _ in range(4):      # 3-channel stem, Wo % 4 == 0
        cases.append(((1, 7, 7), (1, 2, 2), (0, 3, 3), int(rng.integers(1, 4)), 3, int(rng.choice([16, 45, 48])),
                      (int(rng.integers(1, 5)), 2 * int(rng.integers(2, 30)), 8 * int(rng.integers(1, 16)))))
    for k, s, p, n, cin, cout, dims in cases:
        x = torch.randn(n, cin, *dims, generator=g)
        wt = torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * k[0] * k[1] * k[2])
        xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
        yr = F.conv3d(xr, wr, None, stride=s, padding=p)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy.double())
        xg, wg = x.to(DEV).requires_grad_(), wt.to(DEV).requires_grad_()
        tag = f"n{n} c{cin}->{cout} k{k} s{s} p{p} in{dims}"
        yg = ops.conv3d(xg, wg, None, s, p)
        close(yg, yr, rtol=3e-5, what=tag + " fwd")
        yg.backward(dy.to(DEV))
        close(xg.grad, xr.grad, rtol=3e-5, what=tag + " dgrad")
        close(wg.grad, wr.grad, rtol=5e-5, what=tag + " wgrad")


@pytest.mark.parametrize("offset", [0, 1, 3])
def test_relu_bwd_bias_on_misaligned_views(offset):
    """zsv_relu_bwd_bias (C3D's relu(conv + b) backward: ReLU mask and bias gradient in one pass): its float4 path needs
    16-byte aligned bases -- a contiguous view that starts at an odd element of a larger buffer must take the scalar path."""
    import ctypes
    from zeroshotvideoclassification_amd import _lib
    lib = _lib.load()
    n, c, s = 3, 5, 64
    g = torch.Generator().manual_seed(offset)

    def view(t):
        buf = torch.zeros(t.numel() + 8, device=DEV)
        v = buf[offset:offset + t.numel()].view(t.shape)
        v.copy_(t)
        return v

    dy, y = torch.randn(n, c, s, generator=g), torch.randn(n, c, s, generator=g)
    dyd, yd = view(dy), view(y)
    gd = view(torch.zeros(n, c, s))
    assert dyd.is_contiguous() and (dyd.data_ptr() % 16 == 0) == (offset == 0)
    db = torch.empty(c, device=DEV)
    nb = lib.zsv_channel_sum_workspace_bytes(n, c, s)
    ws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=DEV)
    _lib.check(lib.zsv_relu_bwd_bias(dyd.data_ptr(), yd.data_ptr(), gd.data_ptr(), n, c, s, db.data_ptr(), ws.data_ptr(), nb, None),
               "zsv_relu_bwd_bias")
    torch.cuda.synchronize()
    ref = torch.where(y > 0, dy, torch.zeros_like(dy))
    assert torch.equal(gd.cpu(), ref)
    close(db, ref.double().sum(dim=(0, 2)), what="bias gradient")


@pytest.mark.parametrize("n,cin,cout,hw", [(5, 921, 512, (7, 7)), (22, 1152, 512, (7, 7)), (2, 45, 70, (5, 6)), (3, 64, 48, (8, 8))],
                         ids=["T9", "T10_full_batch", "odd_channels", "whole_pieces"])
def test_two_frame_temporal_conv_in_dense_form(n, cin, cout, hw, monkeypatch):
    """Conv3d(k=(3,1,1), pad (1,0,0)) on TWO frames (resnet.py:46-52 at layer4) runs as a dense 1x1x1 convolution over
    (channel, frame) pairs -- 4 instead of 6 products per (co, c, position), csrc/conv_params.h t2_dense_shape.  Forward, input
    gradient and weight gradient against torch CPU fp64 and against the direct kernels (ZSV_NO_T2_DENSE=1)."""
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, 2, *hw, generator=g)
    w = torch.randn(cout, cin, 3, 1, 1, generator=g) / np.sqrt(3 * cin)
    dy = torch.randn(n, cout, 2, *hw, generator=g)

    def run():
        xd = x.to(DEV).requires_grad_()
        wd = w.to(DEV).requires_grad_()
        y = ops.conv3d(xd, wd, None, 1, (1, 0, 0))
        y.backward(dy.to(DEV))
        torch.cuda.synchronize()
        return y.detach(), xd.grad, wd.grad

    dense = run()
    monkeypatch.setenv("ZSV_NO_T2_DENSE", "1")
    direct = run()
    monkeypatch.delenv("ZSV_NO_T2_DENSE")
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    yr = F.conv3d(xr, wr, padding=(1, 0, 0))
    yr.backward(dy.double())
    for got, other, ref, what in zip(dense, direct, (yr, xr.grad, wr.grad), ("forward", "dgrad", "wgrad")):
        close(got, ref, what=f"dense {what}")
        close(got, other.double(), rtol=1e-5, what=f"dense vs direct {what}")
    assert not torch.equal(dense[0], direct[0]), "the dense and the direct form should not be the same kernel path"


def test_weight_gradient_slab_sum_rows_equals_the_generic_sum(monkeypatch):
    """conv_wgrad.hip: few slices of a large gradient are added by slab_sum_rows_kernel (coalesced, LDS-transposed) -- same
    slice order as slab_sum_kernel, so the same bits (ZSV_NO_SLAB_SUM_ROWS=1 selects the generic kernel)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 96, 2, 7, 7, generator=g).to(DEV)
    dy = torch.randn(6, 200, 2, 7, 7, generator=g).to(DEV)
    w = torch.randn(200, 96, 1, 3, 3, generator=g).to(DEV)

    def wgrad():
        xd, wd = x.clone().requires_grad_(), w.clone().requires_grad_()
        ops.conv3d(xd, wd, None, 1, (0, 1, 1)).backward(dy)
        torch.cuda.synchronize()
        return wd.grad

    a = wgrad()
    monkeypatch.setenv("ZSV_NO_SLAB_SUM_ROWS", "1")
    b = wgrad()
    assert torch.equal(a, b)
    ref = torch.nn.grad.conv3d_weight(x.double().cpu(), w.shape, dy.double().cpu(), padding=(0, 1, 1))
    close(a, ref, what="wgrad")


def test_tiled_weight_pack_equals_the_elementwise_pack(monkeypatch):
    """conv_tap.hip: pack_weights_tiled_kernel (coalesced reads and writes through an LDS tile) writes exactly the panel of
    pack_weights_kernel: forward and both gradients of direct-kernel layers are bit-identical with ZSV_NO_PACK_TILED=1."""
    cases = [((4, 128, 4, 14, 14), (300, 128, 1, 3, 3), (1, 2, 2), (0, 1, 1)),       # strided spatial (forward + class-by-class packs)
             ((3, 200, 4, 7, 7), (120, 200, 3, 1, 1), (2, 1, 1), (1, 0, 0)),         # strided temporal
             ((3, 96, 2, 7, 7), (130, 96, 1, 3, 3), 1, (0, 1, 1)),                   # odd width: direct kernel
             ((2, 40, 3, 6, 6), (50, 40, 3, 3, 3), (2, 2, 2), (1, 1, 1))]            # 27 taps, 8 residue classes
    for xs, ws, stride, pad in cases:
        g = torch.Generator().manual_seed(xs[1])
        x, w = torch.randn(*xs, generator=g).to(DEV), (torch.randn(*ws, generator=g) / 10).to(DEV)

        def run():
            xd, wd = x.clone().requires_grad_(), w.clone().requires_grad_()
            y = ops.conv3d(xd, wd, None, stride, pad)
            y.backward(torch.ones_like(y) * 0.5 + y.detach() * 0.1)
            torch.cuda.synchronize()
            return y.detach(), xd.grad, wd.grad

        a = run()
        monkeypatch.setenv("ZSV_NO_PACK_TILED", "1")
        b = run()
        monkeypatch.delenv("ZSV_NO_PACK_TILED")
        for u, v in zip(a, b):
            assert torch.equal(u, v), (xs, ws)
        ref = F.conv3d(x.double().cpu(), w.double().cpu(), stride=stride, padding=pad)
        close(a[0], ref, what=f"forward {ws}")


VW_CASES = [
    # name, N, Cin, Cout, kT, (T, H, W)
    ("layer4_7x7", 22, 512, 921, 1, (2, 7, 7)),            # S9: 2464 virtual voxels, K parts
    ("layer4_dgrad_rows", 22, 921, 512, 1, (2, 7, 7)),
    ("layer3_14x14", 22, 256, 460, 1, (4, 14, 14)),         # S6
    ("w15_small", 3, 32, 40, 1, (3, 9, 15)),               # one real voxel short of the virtual row; partial last tile
    ("w30", 2, 24, 48, 1, (2, 11, 30)),
    ("w62", 2, 16, 32, 1, (1, 9, 62)),
    ("c3d_conv5_like_333", 6, 64, 80, 3, (2, 7, 7)),       # 9 row taps, temporal padding
    ("r3d_layer3_333", 8, 48, 64, 3, (4, 14, 14)),
]


@pytest.mark.parametrize("case", VW_CASES, ids=[c[0] for c in VW_CASES])
def test_conv3d_winograd_virtual_width(case, monkeypatch):
    """Row lengths that are not a multiple of 4 (layer3's 14, layer4's 7, resnet.py:217-220) run the F(4,3) kernel on rows
    padded -- virtually -- to 8 / 16 / 32 / 64 voxels (conv_wino4_kernel VW): forward (plain, + BatchNorm partial statistics, +
    bias + residual + ReLU) and input gradient (plain, + shortcut gradient) against torch CPU fp64 and against the kernels the
    geometry used before (ZSV_WINO_NO_VW=1: F(2,3) for even W, the direct kernel for odd W)."""
    import ctypes
    from zeroshotvideoclassification_amd import _lib
    name, n, cin, cout, kt, (t, h, w) = case
    assert uses_f43(w) and w % 4 != 0
    monkeypatch.setenv("ZSV_WINO_VW_MIN_WGS", "1")          # (small cases: fewer workgroups than the production threshold)
    g = torch.Generator().manual_seed(len(name) * 131 + cin)
    k, pad = (kt, 3, 3), (kt // 2, 1, 1)
    wt = torch.randn(cout, cin, *k, generator=g) / np.sqrt(cin * 9 * kt)
    x = torch.randn(n, cin, t, h, w, generator=g)
    bias = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, t, h, w, generator=g)
    dy = torch.randn(n, cout, t, h, w, generator=g)
    add = torch.randn(n, cin, t, h, w, generator=g)
    lib = _lib.load()
    d = ops.conv_desc(x.shape, wt.shape, 1, pad)
    xd, wd, bd, rd, dyd, addd = (v.to(DEV) for v in (x, wt, bias, res, dy, add))
    big = n * cin * t * h * w * cout > 2e9                  # fp64 CPU reference on two clips only
    clips = [0, n - 1] if big else list(range(n))

    def fwd(mode):
        y = torch.full((n, cout, t, h, w), float("nan"), device=DEV)
        nbytes = lib.zsv_conv3d_fwd_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=DEV)
        stats, tiles = None, 0
        if mode == "stats":
            tiles = lib.zsv_conv3d_fwd_stat_tiles(ctypes.byref(d), y.data_ptr())
            if tiles <= 0:
                return None, None
            stats = torch.full((2, cout, tiles), float("nan"), device=DEV)
        full = mode == "full" and lib.zsv_conv3d_fwd_add_supported(ctypes.byref(d))
        if mode == "full" and not full:
            return None, None
        _lib.check(lib.zsv_conv3d_fwd_full(ctypes.byref(d), xd.data_ptr(), wd.data_ptr(), bd.data_ptr() if full else None,
                                           rd.data_ptr() if full else None, y.data_ptr(), 1 if full else 0,
                                           stats.data_ptr() if stats is not None else None, tiles, ws.data_ptr(), nbytes, None), "fwd")
        torch.cuda.synchronize()
        return y, stats

    def dgrad(with_add):
        if with_add and not lib.zsv_conv3d_dgrad_add_supported(ctypes.byref(d)):
            return None
        dx = torch.full((n, cin, t, h, w), float("nan"), device=DEV)
        nbytes = lib.zsv_conv3d_dgrad_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=DEV)
        _lib.check(lib.zsv_conv3d_dgrad_add(ctypes.byref(d), dyd.data_ptr(), wd.data_ptr(), addd.data_ptr() if with_add else None,
                                            dx.data_ptr(), ws.data_ptr(), nbytes, None), "dgrad")
        torch.cuda.synchronize()
        return dx

    ref = {i: F.conv3d(x[i:i + 1].double(), wt.double(), padding=pad) for i in clips}
    y, _ = fwd("plain")
    assert torch.isfinite(y).all()                          # every real voxel was written (the buffer started as NaN)
    for i in clips:
        close(y[i:i + 1], ref[i], what=f"forward clip {i}")
    assert torch.equal(y, fwd("plain")[0]), "bitwise reproducible"
    ys, stats = fwd("stats")
    if ys is not None:                                      # (K parts have no fused statistics)
        assert torch.equal(ys, y)
        if not big:
            full_ref = torch.cat([ref[i] for i in clips])
            close(stats[0].double().sum(1), full_ref.sum(dim=(0, 2, 3, 4)), rtol=1e-4, what="sum y")
            close(stats[1].double().sum(1), (full_ref * full_ref).sum(dim=(0, 2, 3, 4)), rtol=1e-4, what="sum y^2")
    yf, _ = fwd("full")
    if yf is not None:
        for i in clips:
            close(yf[i:i + 1], torch.relu(ref[i] + bias.double().view(1, -1, 1, 1, 1) + res[i:i + 1].double()), what="bias + residual + relu")
    dx = dgrad(False)
    assert torch.isfinite(dx).all()
    for i in clips:
        close(dx[i:i + 1], torch.nn.grad.conv3d_input((1, cin, t, h, w), wt.double(), dy[i:i + 1].double(), padding=pad), what=f"dgrad clip {i}")
    dxa = dgrad(True)
    if dxa is not None:
        close(dxa, dx.double() + add.double().to(DEV), rtol=1e-6, what="dgrad + shortcut gradient")
    monkeypatch.setenv("ZSV_WINO_NO_VW", "1")
    y2, dx2 = fwd("plain")[0], dgrad(False)
    monkeypatch.delenv("ZSV_WINO_NO_VW")
    close(y, y2.double(), rtol=1e-5, what="virtual width vs previous kernel, forward")
    close(dx, dx2.double(), rtol=1e-5, what="virtual width vs previous kernel, dgrad")
    assert not torch.equal(y, y2) and not torch.equal(dx, dx2), "the two paths should not be the same kernel"


PANEL_CASES = [
    # name, x shape, w shape, stride, padding
    ("winograd_f43", (3, 64, 4, 24, 24), (144, 64, 1, 3, 3), 1, (0, 1, 1)),
    ("virtual_width_7", (6, 96, 2, 7, 7), (200, 96, 1, 3, 3), 1, (0, 1, 1)),
    ("temporal_winograd", (3, 144, 8, 16, 16), (64, 144, 3, 1, 1), 1, (1, 0, 0)),
    ("two_frames_dense", (5, 130, 2, 7, 7), (70, 130, 3, 1, 1), 1, (1, 0, 0)),
    ("strided_spatial", (3, 64, 4, 28, 28), (230, 64, 1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("strided_temporal", (3, 230, 8, 14, 14), (128, 230, 3, 1, 1), (2, 1, 1), (1, 0, 0)),
    ("pointwise", (4, 64, 4, 14, 14), (128, 64, 1, 1, 1), 1, (0, 0, 0)),
    ("full_333", (2, 32, 6, 12, 16), (48, 32, 3, 3, 3), 1, (1, 1, 1)),
]


@pytest.mark.parametrize("case", PANEL_CASES, ids=[c[0] for c in PANEL_CASES])
def test_weight_panels_packed_ahead_of_the_call(case):
    """C ABI: zsv_conv3d_panel_query / _panel_job / zsv_pack_multi / zsv_conv3d_*_panel.  The panel packed by the multi-job launch
    from the call's own job, fed to the *_panel entry point, gives the bits of the ordinary entry point (which packs per call):
    forward (plain and with a bias + ReLU) and input gradient; several jobs share one table / one launch."""
    import ctypes
    from zeroshotvideoclassification_amd import _lib
    name, xs, ws_, stride, pad = case
    lib = _lib.load()
    g = torch.Generator().manual_seed(len(name))
    x = torch.randn(*xs, generator=g).to(DEV)
    w = (torch.randn(*ws_, generator=g) / np.sqrt(np.prod(ws_[1:]))).to(DEV)
    bias = torch.randn(ws_[0], generator=g).to(DEV)
    d = ops.conv_desc(xs, ws_, stride, pad)
    y_shape = (d.N, d.Cout, d.To, d.Ho, d.Wo)
    dy = torch.randn(*y_shape, generator=g).to(DEV)
    variants = [(0, 0), (0, 1), (1, 0)]                       # (direction, extras)
    jobs, panels = [], {}
    for direction, extras in variants:
        nb = ctypes.c_size_t(0)
        _lib.check(lib.zsv_conv3d_panel_query(ctypes.byref(d), direction, extras, ctypes.byref(nb)), "query")
        assert nb.value > 0, (name, direction, extras)
        panel = torch.full((nb.value,), 0xFF, dtype=torch.uint8, device=DEV)
        job = _lib.PackJob()
        _lib.check(lib.zsv_conv3d_panel_job(ctypes.byref(d), direction, extras, w.data_ptr(), panel.data_ptr(), nb.value, ctypes.byref(job)), "job")
        assert job.total * 4 <= nb.value and job.w == w.data_ptr() and job.out == panel.data_ptr()
        jobs.append(job)
        panels[(direction, extras)] = panel
    first, raw = 0, bytearray()
    for job in jobs:
        job.first_block = first
        first += (job.total + 1023) // 1024
        raw += bytes(job)
    table = torch.frombuffer(raw, dtype=torch.uint8).to(DEV)
    _lib.check(lib.zsv_pack_multi(table.data_ptr(), len(jobs), first, None), "pack_multi")
    nf, nd = lib.zsv_conv3d_fwd_workspace_bytes(ctypes.byref(d)), lib.zsv_conv3d_dgrad_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(int(nf), int(nd), 16), dtype=torch.uint8, device=DEV)
    for extras in (0, 1):
        y_ref, y = torch.empty(y_shape, device=DEV), torch.empty(y_shape, device=DEV)
        b = bias.data_ptr() if extras else None
        _lib.check(lib.zsv_conv3d_fwd_full(ctypes.byref(d), x.data_ptr(), w.data_ptr(), b, None, y_ref.data_ptr(), extras, None, 0, ws.data_ptr(), nf, None), "fwd")
        p = panels[(0, extras)]
        _lib.check(lib.zsv_conv3d_fwd_full_panel(ctypes.byref(d), x.data_ptr(), w.data_ptr(), b, None, y.data_ptr(), extras, None, 0, ws.data_ptr(), nf, None,
                                                 p.data_ptr(), p.numel()), "fwd panel")
        torch.cuda.synchronize()
        assert torch.equal(y, y_ref), (name, "forward", extras)
    dx_ref, dx = torch.empty(xs, device=DEV), torch.empty(xs, device=DEV)
    _lib.check(lib.zsv_conv3d_dgrad(ctypes.byref(d), dy.data_ptr(), w.data_ptr(), dx_ref.data_ptr(), ws.data_ptr(), nd, None), "dgrad")
    p = panels[(1, 0)]
    _lib.check(lib.zsv_conv3d_dgrad_add_panel(ctypes.byref(d), dy.data_ptr(), w.data_ptr(), None, dx.data_ptr(), ws.data_ptr(), nd, None, p.data_ptr(), p.numel()),
               "dgrad panel")
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref), (name, "dgrad")
    ref = F.conv3d(x.double().cpu(), w.double().cpu(), stride=stride, padding=pad)
    close(y, F.relu(ref + bias.double().cpu().view(1, -1, 1, 1, 1)), what="forward + bias + relu")
    # a too-small panel is refused, a stem has no panel
    assert lib.zsv_conv3d_fwd_full_panel(ctypes.byref(d), x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), 0, None, 0, ws.data_ptr(), nf, None,
                                         panels[(0, 0)].data_ptr(), 16) != 0
    ds = ops.conv_desc((2, 3, 4, 32, 32), (45, 3, 1, 7, 7), (1, 2, 2), (0, 3, 3))
    nb = ctypes.c_size_t(1)
    _lib.check(lib.zsv_conv3d_panel_query(ctypes.byref(ds), 0, 0, ctypes.byref(nb)), "query stem")
    assert nb.value == 0


@pytest.mark.parametrize("xs,cout,k", [((4, 64, 8, 28, 28), 144, (1, 3, 3)), ((2, 64, 4, 14, 14), 64, (3, 3, 3)),
                                       ((4, 32, 4, 16, 16), 48, (1, 3, 3)), ((2, 144, 4, 12, 12), 64, (3, 1, 1)), ((4, 144, 8, 28, 28), 64, (3, 1, 1)),
                                       ((3, 64, 8, 28, 30), 230, (1, 3, 3))])
def test_wgrad_with_a_kept_tap_validity_table_is_the_same_bits(xs, cout, k, monkeypatch):
    """zsv_conv3d_wgrad_masked: the per-voxel tap-validity table of the stride-1 weight-gradient kernels depends on the geometry
    only; a caller-kept table (ops._wgrad_mask: one per geometry and stream) gives the same bits as the table rebuilt per call,
    and a geometry whose kernel reads none reports 0 bytes."""
    from ctypes import byref, c_void_p
    from zeroshotvideoclassification_amd import _lib
    lib = _lib.load()
    n, cin, t, h, w = xs
    pad = tuple((v - 1) // 2 for v in k)
    g = torch.Generator().manual_seed(zlib.crc32(repr((xs, cout, k)).encode()))
    x = torch.randn(xs, generator=g).to(DEV)
    wshape = (cout, cin) + tuple(k)
    d = ops.conv_desc(x.shape, wshape, 1, pad)
    dy = torch.randn((n, cout, d.To, d.Ho, d.Wo), generator=g).to(DEV)
    nbytes = lib.zsv_conv3d_wgrad_workspace_bytes(byref(d))
    stream = c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(mask):
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=DEV)
        dw = torch.empty(wshape, device=DEV)
        _lib.check(lib.zsv_conv3d_wgrad_masked(byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nbytes,
                                               None if mask is None else mask.data_ptr(), stream), "zsv_conv3d_wgrad_masked")
        return dw

    plain = torch.empty(wshape, device=DEV)
    ws0 = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=DEV)
    _lib.check(lib.zsv_conv3d_wgrad(byref(d), x.data_ptr(), dy.data_ptr(), plain.data_ptr(), ws0.data_ptr(), nbytes, stream), "zsv_conv3d_wgrad")
    mbytes = int(lib.zsv_conv3d_wgrad_mask_bytes(byref(d)))
    assert torch.equal(run(None), plain)
    if k == (3, 1, 1) and n * t * h * w >= 16384:
        assert mbytes == 0                                   # the frame-ring / Winograd-along-T kernels read no table
    if mbytes == 0:
        assert ops._wgrad_mask(d, x.device, stream) is None
    else:
        assert mbytes == 4 * t * h * w
        mask = ops._wgrad_mask(d, x.device, stream)
        assert mask is not None and mask.numel() == mbytes and ops._wgrad_mask(d, x.device, stream) is mask      # kept
        assert torch.equal(run(mask), plain)
    # the autograd path uses the kept table; switching the cache off gives the same gradient
    xg = x.clone()
    wt = (torch.randn(wshape, generator=g) * 0.05).to(DEV).requires_grad_(True)
    ops.conv3d(xg, wt, None, 1, pad).backward(dy)
    ops.join_wgrad_streams()
    g1 = wt.grad.clone()
    wt.grad = None
    monkeypatch.setenv("ZSV_NO_WGRAD_MASK_CACHE", "1")
    ops.conv3d(xg, wt, None, 1, pad).backward(dy)
    ops.join_wgrad_streams()
    torch.cuda.synchronize()
    assert torch.equal(g1, wt.grad)
