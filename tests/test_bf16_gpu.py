"""bf16 inference path (BASELINE config 5): zsv_conv3d_bf16_fwd per layer against an fp64 CPU
convolution of the SAME bf16-rounded operands, and the whole Bf16Engine against the fp32 oracle
fixtures.

Tolerances (written here because north_star's 1e-3 is the fp32 figure): a single layer differs from
the exact result of its rounded operands only by fp32 accumulation order and the final rounding to
bf16 (half an ulp = 2^-9 relative), so 2^-8 relative + 1e-3 absolute per element; end to end the
bf16 activations of ~40 layers compound, and the 300-d unit-norm embedding is required to stay
within 2e-2 (max abs, components are O(0.06)) and cosine >= 0.999 of the fp32 oracle's.
"""
import numpy as np
import pytest
import zlib

import torch
import torch.nn.functional as F

from helpers import load_golden, make_opt
from zeroshotvideoclassification_amd import inference, network, ops, synthetic, train

pytestmark = pytest.mark.gpu
DEV = "cuda"


def to_ndhwc(x, pitch):
    """(N,C,T,H,W) fp32 -> [N][T][H][W][pitch] bf16 (test-side layout change)."""
    n, c, t, h, w = x.shape
    out = torch.zeros((n, t, h, w, pitch), dtype=torch.bfloat16, device=x.device)
    out[..., :c] = x.permute(0, 2, 3, 4, 1).to(torch.bfloat16)
    return out


def from_ndhwc(y, c):
    return y[..., :c].permute(0, 4, 1, 2, 3).float()


def bf16_round(t):
    return t.to(torch.bfloat16).float()


CASES = [
    # n, cin, cout, (t,h,w), kernel, stride, padding, residual, relu
    (2, 64, 144, (4, 14, 14), (1, 3, 3), (1, 1, 1), (0, 1, 1), False, True),     # S1-like, row tile 144 + zero-filled pitch
    (2, 144, 64, (4, 14, 14), (3, 1, 1), (1, 1, 1), (1, 0, 0), True, True),      # T1-like + residual, K pitch 160
    (1, 64, 230, (4, 16, 16), (1, 3, 3), (1, 2, 2), (0, 1, 1), False, True),     # strided spatial
    (1, 230, 128, (6, 8, 8), (3, 1, 1), (2, 1, 1), (1, 0, 0), False, False),     # strided temporal, no relu
    (3, 64, 128, (4, 12, 12), (1, 1, 1), (2, 2, 2), (0, 0, 0), False, False),    # shortcut 1x1x1
    (1, 45, 64, (5, 9, 11), (3, 1, 1), (1, 1, 1), (1, 0, 0), False, True),       # T0: 45 -> pitch 64
    (1, 128, 288, (3, 10, 10), (3, 3, 3), (1, 1, 1), (1, 1, 1), True, True),     # r3d-style 3x3x3
    (1, 512, 1152, (2, 7, 7), (1, 3, 3), (1, 1, 1), (0, 1, 1), False, True),     # layer4 width
    (1, 921, 512, (2, 4, 4), (3, 1, 1), (1, 1, 1), (1, 0, 0), True, True),       # 921 -> pitch 928, ragged voxel tile
    (5, 32, 33, (1, 3, 3), (1, 3, 3), (1, 1, 1), (0, 1, 1), False, False),       # tiny, Cout just over one chunk
    (2, 144, 64, (8, 8, 8), (3, 1, 1), (1, 1, 1), (1, 0, 0), True, True),        # temporal, frames-x-positions tiles (8 x 32)
    (1, 45, 64, (16, 8, 12), (3, 1, 1), (1, 1, 1), (1, 0, 0), False, True),      # same, two frame blocks per clip
    (8, 32, 128, (16, 28, 28), (3, 1, 1), (1, 1, 1), (1, 0, 0), True, False),    # 16 frames x 16 positions, 128-row tiles
]


@pytest.mark.parametrize("case", CASES, ids=[f"c{i}" for i in range(len(CASES))])
def test_conv_bf16_matches_fp64_of_rounded_operands(case):
    n, cin, cout, (t, h, w), k, s, p, use_res, relu = case
    g = torch.Generator().manual_seed(zlib.crc32(str(case).encode()))   # (not hash(): randomised per process)
    x = bf16_round(torch.randn((n, cin, t, h, w), generator=g))
    wgt = torch.randn((cout, cin) + k, generator=g) / np.sqrt(cin * np.prod(k))
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ref = F.conv3d(x.double(), bf16_round(wgt * scale.view(-1, 1, 1, 1, 1)).double(), stride=s, padding=p)
    ref = ref + shift.double().view(1, -1, 1, 1, 1)
    res = None
    if use_res:
        res = bf16_round(torch.randn(ref.shape, generator=g))
        ref = ref + res.double()
    if relu:
        ref = ref.clamp_min(0)

    d = ops.conv_desc(x.shape, wgt.shape, s, p)
    blob = inference.pack_conv(d, wgt.to(DEV), scale.to(DEV), shift.to(DEV))
    xb = to_ndhwc(x.to(DEV), inference.channel_pitch(cin))
    rb = to_ndhwc(res.to(DEV), inference.channel_pitch(cout)) if use_res else None
    y = inference.conv_bf16(d, xb, blob, rb, relu)
    assert y.shape[-1] == inference.channel_pitch(cout)
    assert torch.count_nonzero(y[..., cout:]) == 0, "pad channels must be written as zero"
    got = from_ndhwc(y, cout).cpu().double()
    err = (got - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + 1e-3
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e}, worst ratio {(err / tol).max().item():.2f}"


NINE_TAP_CASES = [
    # n, cin, cout, (t,h,w), kernel: geometries that reach conv_bf16_same9_kernel (one LDS image for the nine (kh, kw) taps)
    (3, 144, 64, (16, 56, 56), (1, 3, 3)),      # 64-row tile, W = 56 (24-piece image): the input gradient of layer1's spatial convolutions
    (8, 128, 128, (16, 28, 28), (1, 3, 3)),     # one 128-row tile, W = 28 (20-piece image): layer2's input gradients
    (2, 128, 144, (6, 20, 28), (3, 3, 3)),      # 3x3x3 taps (C3D / R3D-18), 144-row tile, ragged last voxel tile
    (1, 160, 288, (4, 9, 7), (3, 3, 3)),        # 3x3x3, two row tiles, W = 7: a 256-voxel tile spans 36 image rows
]


@pytest.mark.parametrize("case", NINE_TAP_CASES, ids=[f"n{i}" for i in range(len(NINE_TAP_CASES))])
def test_conv_bf16_nine_tap_image(case, monkeypatch):
    """conv_bf16_same9_kernel against a CPU convolution of the same bf16-rounded operands (fp32 accumulation: these are too big
    for an fp64 reference in seconds) and against the per-(kt, kh) image kernel it replaces (ZSV_BF16_NO_SAME9=1)."""
    n, cin, cout, (t, h, w), k = case
    p = (k[0] // 2, 1, 1)
    g = torch.Generator().manual_seed(zlib.crc32(str(case).encode()))
    x = bf16_round(torch.randn((n, cin, t, h, w), generator=g))
    wgt = torch.randn((cout, cin) + k, generator=g) / np.sqrt(cin * np.prod(k))
    shift = torch.randn(cout, generator=g) * 0.1
    ref = F.conv3d(x, bf16_round(wgt), stride=1, padding=p) + shift.view(1, -1, 1, 1, 1)
    d = ops.conv_desc(x.shape, wgt.shape, (1, 1, 1), p)
    xb = to_ndhwc(x.to(DEV), inference.channel_pitch(cin))

    def run():
        blob = inference.pack_conv(d, wgt.to(DEV), None, shift.to(DEV))
        return from_ndhwc(inference.conv_bf16(d, xb, blob, None, False), cout).cpu()

    got = run()
    err = (got - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + 2e-3
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e}, worst ratio {(err / tol).max().item():.2f}"
    monkeypatch.setenv("ZSV_BF16_NO_SAME9", "1")
    other = run()
    # (with the chunk-outer walk -- three or more K chunks -- both kernels add the taps in the same order: identical bits are legitimate)
    err = (got - other).abs()
    assert bool((err <= ref.abs() * 2.0 ** -7 + 2e-3).all()), f"nine-tap image vs per-(kt, kh) images: max diff {err.max().item():.3e}"


@pytest.mark.parametrize("kernel,stride,padding", [((1, 7, 7), (1, 2, 2), (0, 3, 3)), ((3, 7, 7), (1, 2, 2), (1, 3, 3))])
def test_clip_convolution_folded_form(kernel, stride, padding):
    """The stems (resnet.py:170,181): 3 channels, border materialised, kw folded into K."""
    g = torch.Generator().manual_seed(11)
    n, t, h, w, cout = 2, 4, 20, 24, 45
    x = bf16_round(torch.rand((n, 3, t, h, w), generator=g) * 0.5 - 0.5)
    wgt = torch.randn((cout, 3) + kernel, generator=g) / np.sqrt(3 * np.prod(kernel))
    conv = torch.nn.Conv3d(3, cout, kernel, stride, padding, bias=False)
    with torch.no_grad():
        conv.weight.copy_(wgt)
    bn = torch.nn.BatchNorm3d(cout)
    with torch.no_grad():
        bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(cout, generator=g) * 0.1)
    op = inference._ConvOp(conv.to(DEV), bn.to(DEV), relu=True)
    pad_h, pad_w, hp, wp = op.input_border(h, w)
    xb = inference.clip_to_bf16(x.to(DEV), pad_h, pad_w, hp, wp)
    assert xb.shape == (n, t, hp, wp, 4)
    wo = (w + 2 * padding[2] - kernel[2]) // stride[2] + 1
    y = op(xb, wo=wo)
    scale, shift = inference.fold_bn(bn, conv)
    ref = F.conv3d(x.double(), bf16_round(wgt * scale.cpu().view(-1, 1, 1, 1, 1)).double(), stride=stride, padding=padding)
    ref = (ref + shift.cpu().double().view(1, -1, 1, 1, 1)).clamp_min(0)
    got = from_ndhwc(y, cout).cpu().double()
    assert got.shape == ref.shape
    err = (got - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -8 + 1e-3).all()), f"max err {err.max().item():.3e}"


def test_meanpool_bf16():
    x = torch.randn((3, 2, 5, 7, 96), device=DEV).to(torch.bfloat16)
    got = inference.meanpool_bf16(x, 70)
    ref = x.float().mean(dim=(1, 2, 3))[:, :70]
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6)


def test_rejects():
    x = torch.zeros((1, 2, 4, 4, 64), dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        inference.meanpool_bf16(x, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        inference.Bf16Engine(network.get_network(make_opt("r2plus1d_18")))
    d = ops.conv_desc((1, 64, 2, 4, 4), (64, 64, 1, 3, 3), 1, (0, 1, 1))
    blob = inference.pack_conv(d, torch.zeros((64, 64, 1, 3, 3), device=DEV), None, None)
    with pytest.raises(RuntimeError, match="does not match"):
        inference.conv_bf16(d, torch.zeros((1, 2, 4, 4, 32), dtype=torch.bfloat16, device=DEV), blob)


def _model(name, seed, jitter=True):
    if name == "mc3_18":            # not reachable through get_network's dispatch (network.py:24-44)
        from zeroshotvideoclassification_amd import resnet
        model = network.Model(resnet.mc3_18)
    else:
        model = network.get_network(make_opt(name))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=seed, bn_jitter=jitter))
    return model.to(DEV).eval()


@pytest.mark.parametrize("name", ["r2plus1d_18", "r3d_18", "mc3_18"])
def test_engine_matches_fp32_eval_forward(name):
    """Same weights, eval mode: bf16 engine vs the fp32 HIP path (itself pinned to the oracle)."""
    model = _model(name, seed=7)
    x = synthetic.synthetic_clips(3, 8, 64).to(DEV)
    with torch.no_grad():
        ref, _ = model(x)
    emb, second = inference.Bf16Engine(model)(x)
    assert second is None and emb.shape == ref.shape and emb.dtype == torch.float32
    assert torch.allclose(emb.norm(dim=1), torch.ones(3, device=DEV), atol=1e-5)
    cos = (emb * ref).sum(dim=1)
    assert cos.min().item() >= 0.999, cos
    assert (emb - ref).abs().max().item() <= 2e-2


def test_engine_matches_oracle_fixture_t32():
    """Config E fixture: 32-frame eval-mode embedding from the fp32 CPU oracle (tests/golden)."""
    from test_model_gpu import build
    g, model, _ = build("r2plus1d_jitter")
    model.eval()
    x32 = synthetic.synthetic_clips(1, 32, int(g["meta_size"]), seed=99).to(DEV)
    emb, _ = inference.Bf16Engine(model)(x32)
    ref = torch.from_numpy(g["emb_eval_t32_f32"]).to(DEV)
    cos = (emb * ref).sum(dim=1)
    assert cos.min().item() >= 0.999, cos
    assert (emb - ref).abs().max().item() <= 2e-2


def test_engines_match_the_reference_fixture_t32_batch_of_4():
    """Config E at a batch size > 1: the reference's own eval-mode embeddings of FOUR 32-frame clips
    (tests/golden/r2plus1d_t32_batch.npz, written by oracle/make_golden.py::run_t32_batch from the imported network.py).
    bf16 engine: 2e-2 absolute / cosine >= 0.999 per clip; folded fp32 engine and the module forward: 1e-4 relative."""
    from helpers import load_golden, make_opt
    from zeroshotvideoclassification_amd import network
    g = load_golden("r2plus1d_t32_batch")
    model = network.get_network(make_opt(str(g["meta_network"])))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0, bn_jitter=bool(g["meta_bn_jitter"])))
    model.to(DEV).eval()
    x = synthetic.synthetic_clips(int(g["meta_n"]), int(g["meta_frames"]), int(g["meta_size"]), seed=int(g["meta_seed"])).to(DEV)
    ref = torch.from_numpy(g["emb_eval_t32_f32"]).to(DEV)
    assert ref.shape == (4, 300)
    emb, _ = inference.Bf16Engine(model)(x)
    cos = (emb * ref).sum(dim=1)
    assert cos.min().item() >= 0.999, cos
    assert (emb - ref).abs().max().item() <= 2e-2
    with torch.no_grad():
        plain, _ = model(x)
    folded, _ = inference.Fp32Engine(model)(x)
    scale = ref.abs().max().item()
    assert (plain - ref).abs().max().item() <= 1e-4 * scale
    assert (folded - ref).abs().max().item() <= 1e-4 * scale


def test_engine_ranking_agrees_with_fp32():
    """Nearest-class ranking (main.py:316-325) from bf16 embeddings vs fp32 embeddings."""
    model = _model("r2plus1d_18", seed=3)
    x = synthetic.synthetic_clips(8, 8, 64, seed=99).to(DEV)
    table = synthetic.class_table(101).to(DEV)
    with torch.no_grad():
        ref, _ = model(x)
    emb, _ = inference.Bf16Engine(model)(x)
    top_ref = train.cosine_ranking(ref, table)[:, :5]
    top = train.cosine_ranking(emb, table)[:, :5]
    assert (top[:, 0] == top_ref[:, 0]).float().mean().item() >= 0.75
    overlap = np.mean([len(set(a.tolist()) & set(b.tolist())) for a, b in zip(top, top_ref)])
    assert overlap >= 4.0


def test_evaluate_protocol_in_bf16():
    """train.evaluate(dtype=bf16) = the fp32 protocol (main.py:224-325) with the engine's forward."""
    model = _model("r2plus1d_18", seed=5)
    table = synthetic.class_table(51)
    batches = []
    for i in range(2):
        x = synthetic.synthetic_clips(4, 8, 64, seed=300 + i)
        labels, z = synthetic.synthetic_targets(4, 51, rank=i)
        batches.append((x, labels, z))
    a = train.evaluate(model, batches, table, device=torch.device(DEV), splits=2)
    b = train.evaluate(model, batches, table, device=torch.device(DEV), splits=2, dtype=torch.bfloat16)
    assert a["n"] == b["n"] == 8
    eng = inference.engine_for(model)
    assert inference.engine_for(model) is eng, "engine is cached while the weights are unchanged"
    with torch.no_grad():
        model.model.stem[0].weight.mul_(1.0)
    assert inference.engine_for(model) is not eng, "a write to a trunk parameter invalidates it"
    assert abs(a["accuracy_top5"] - b["accuracy_top5"]) <= 12.5 + 1e-6         # at most one of 8 clips flips
    with pytest.raises(RuntimeError, match="not supported"):
        train.evaluate(model, batches, table, device=torch.device(DEV), dtype=torch.float16)


def test_conv_bf16_random_geometries():
    """Seeded sweep over the three bf16 convolution kernels' domains (per-tap, shared image over kw,
    frames-x-positions tiles), strides, channel pitches and ragged tiles."""
    rng = np.random.RandomState(77)
    kernels = [((1, 3, 3), (0, 1, 1)), ((3, 1, 1), (1, 0, 0)), ((3, 3, 3), (1, 1, 1)), ((1, 1, 1), (0, 0, 0))]
    for it in range(20):
        k, p = kernels[rng.randint(len(kernels))]
        s = tuple(int(v) for v in (rng.choice([1, 1, 2]) if k[0] > 1 or k == (1, 1, 1) else 1,
                                   rng.choice([1, 1, 2]) if k[1] > 1 or k == (1, 1, 1) else 1,
                                   rng.choice([1, 1, 2]) if k[2] > 1 or k == (1, 1, 1) else 1))
        cin = int(rng.choice([32, 45, 64, 100, 144]))
        cout = int(rng.choice([33, 64, 128, 144, 230]))
        t, h, w = int(rng.choice([2, 4, 8, 16])), int(rng.choice([4, 7, 8, 12])), int(rng.choice([4, 8, 9, 16]))
        n = int(rng.randint(1, 4))
        use_res, relu = bool(rng.randint(2)), bool(rng.randint(2))
        g = torch.Generator().manual_seed(500 + it)
        x = bf16_round(torch.randn((n, cin, t, h, w), generator=g))
        wgt = torch.randn((cout, cin) + k, generator=g) / np.sqrt(cin * np.prod(k))
        scale = torch.rand(cout, generator=g) + 0.5
        shift = torch.randn(cout, generator=g) * 0.1
        ref = F.conv3d(x.double(), bf16_round(wgt * scale.view(-1, 1, 1, 1, 1)).double(), stride=s, padding=p)
        ref = ref + shift.double().view(1, -1, 1, 1, 1)
        res = None
        if use_res:
            res = bf16_round(torch.randn(ref.shape, generator=g))
            ref = ref + res.double()
        if relu:
            ref = ref.clamp_min(0)
        d = ops.conv_desc(x.shape, wgt.shape, s, p)
        blob = inference.pack_conv(d, wgt.to(DEV), scale.to(DEV), shift.to(DEV))
        y = inference.conv_bf16(d, to_ndhwc(x.to(DEV), inference.channel_pitch(cin)), blob,
                                to_ndhwc(res.to(DEV), inference.channel_pitch(cout)) if use_res else None, relu)
        assert torch.count_nonzero(y[..., cout:]) == 0
        got = from_ndhwc(y, cout).cpu().double()
        err = (got - ref).abs()
        tol = ref.abs() * 2.0 ** -8 + 1e-3
        assert bool((err <= tol).all()), f"it={it} n={n} {cin}->{cout} thw={(t, h, w)} k={k} s={s}: max err {err.max().item():.3e}"


@pytest.mark.parametrize("name", ["r2plus1d_18", "r3d_18"])
def test_fp32_engine_with_folded_batchnorm_matches_module_forward(name):
    """Fp32Engine (BatchNorm folded into the weights, ReLU / residual in the conv epilogue) against the
    module's own eval forward: fp32 both, so only the folding's rounding differs (<= 1e-5 of the scale;
    north_star's bar is 1e-3)."""
    model = _model(name, seed=11)
    x = synthetic.synthetic_clips(3, 8, 64, seed=5).to(DEV)
    with torch.no_grad():
        ref, _ = model(x)
    emb, second = inference.Fp32Engine(model)(x)
    assert second is None and emb.dtype == torch.float32 and emb.shape == ref.shape
    assert (emb - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-7
    table = synthetic.class_table(51)
    labels, z = synthetic.synthetic_targets(3, 51)
    a = train.evaluate(model, [(x.cpu(), labels, z)], table, device=torch.device(DEV), splits=0)
    b = train.evaluate(model, [(x.cpu(), labels, z)], table, device=torch.device(DEV), splits=0, dtype=torch.float32)
    assert a["accuracy"] == b["accuracy"] and a["accuracy_top5"] == b["accuracy_top5"]


@pytest.mark.parametrize("n,t,s", [(1, 24, 96), (3, 8, 80), (2, 16, 128)])
def test_engines_on_other_clip_sizes(n, t, s):
    """Frame counts / resolutions other than the benchmark's: tile-eligibility rules (frames-x-positions
    tiles need T % 8 == 0 and HW % 32 == 0, the shared image stride 1, ...) must fall back, not break."""
    model = _model("r2plus1d_18", seed=2)
    x = synthetic.synthetic_clips(n, t, s, seed=t + s).to(DEV)
    with torch.no_grad():
        ref, _ = model(x)
    b, _ = inference.Bf16Engine(model)(x)
    f, _ = inference.Fp32Engine(model)(x)
    assert (b * ref).sum(dim=1).min().item() >= 0.999 and (b - ref).abs().max().item() <= 2e-2
    assert (f - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-7


@pytest.mark.parametrize("shape,kernel,pad", [((2, 64, 4, 12, 12), (1, 2, 2), (0, 0, 0)), ((1, 128, 4, 6, 6), (2, 2, 2), (0, 0, 0)),
                                              ((3, 512, 2, 7, 7), (2, 2, 2), (0, 1, 1)), ((2, 45, 3, 5, 9), (1, 2, 2), (0, 1, 0))])
def test_maxpool3d_channels_last_bf16(shape, kernel, pad):
    """zsv_maxpool3d_bf16 (network.py:148-163 on the channels-last bf16 layout) against torch's MaxPool3d on the same bf16 values:
    exact (the maximum of bf16 values is one of them)."""
    from zeroshotvideoclassification_amd import amp
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(torch.bfloat16).float()
    ref = F.max_pool3d(x, kernel, kernel, pad)
    y = inference.maxpool3d_bf16(amp.ncdhw_to_cl_bf16(x.to(DEV)), shape[1], kernel, pad)
    torch.cuda.synchronize()
    assert tuple(y.shape[:4]) == (shape[0],) + tuple(ref.shape[2:])
    assert torch.equal(amp.cl_to_ncdhw_f32(y, shape[1]).cpu(), ref)
    assert float(y[..., shape[1]:].float().abs().sum()) == 0.0


def test_c3d_bf16_engine_against_the_reference_fixture_and_the_fp32_path():
    """network.C3D in eval mode on the bf16 engine (inference.Bf16EngineC3D: eight convolutions + five channels-last max-pools in
    bf16, fc6 / regressor fp32): against the reference's own eval-mode embedding (tests/golden/c3d_eval.npz, N = 1) and against
    the fp32 HIP forward on a 2x2-clip batch (bs = 2, nc = 2: the clip mean of network.py:174-176): cosine >= 0.999, 2e-2 absolute.
    `with amp.autocast():` around an eval forward and `train.evaluate(dtype=bfloat16)` take the same route."""
    from helpers import load_golden, make_opt
    from zeroshotvideoclassification_amd import amp, network, train
    g = load_golden("c3d_eval")
    model = network.get_network(make_opt("c3d"))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=0, bn_jitter=bool(g["meta_bn_jitter"])))
    model.to(DEV).eval()
    x = synthetic.synthetic_clips(int(g["meta_n"]), int(g["meta_frames"]), int(g["meta_size"])).to(DEV)
    ref = torch.from_numpy(g["emb_eval_f32"]).to(DEV)
    engine = inference.Bf16EngineC3D(model)
    emb = engine(x)
    assert emb.shape == ref.shape and emb.dtype == torch.float32
    cos = (emb * ref).sum(dim=1)
    assert cos.min().item() >= 0.999, cos
    assert (emb - ref).abs().max().item() <= 2e-2
    xb = torch.stack([synthetic.synthetic_clips(2, 16, 112, seed=31 + i)[:, 0] for i in range(2)]).to(DEV)     # (bs=2, nc=2, 3, 16, 112, 112)
    with torch.no_grad():
        plain = model(xb)
        with amp.autocast():
            routed = model(xb)
    assert torch.equal(routed, inference.engine_for(model, torch.bfloat16)(xb))
    cos = (routed * plain).sum(dim=1)
    assert cos.min().item() >= 0.999, cos
    assert (routed - plain).abs().max().item() <= 2e-2
    table = synthetic.class_table(51, seed=5)
    labels, z = synthetic.synthetic_targets(2, 51, seed=5)
    a = train.evaluate(model, [(xb, labels, z)], table, device=torch.device(DEV), dtype=torch.bfloat16, splits=0)
    b = train.evaluate(model, [(xb, labels, z)], table, device=torch.device(DEV), splits=0)
    assert a["n"] == b["n"] == 2
    with pytest.raises(RuntimeError):
        inference.engine_for(model, torch.float32)
