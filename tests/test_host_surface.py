"""CPU-side checks: the drop-in surface, the C-ABI library, and the no-fallback rule."""
import inspect
import os
import re

import pytest
import torch

from helpers import load_golden, make_opt
from zeroshotvideoclassification_amd import _lib, layers, network, ops, resnet, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "zsv_hip.h")).read()
    declared = set(re.findall(r"\b(zsv_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 24
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/zsv_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes signatures out of sync with the header"
    assert b"gfx950" in lib.zsv_version()
    assert lib.zsv_status_string(0) == b"ok"
    assert b"workspace" in lib.zsv_status_string(3)


def test_workspace_queries_need_no_gpu():
    lib = _lib.load()
    import ctypes
    d = _lib.ConvDesc(22, 64, 16, 56, 56, 144, 16, 56, 56, 1, 3, 3, 1, 1, 1, 0, 1, 1)
    assert lib.zsv_conv3d_wgrad_workspace_bytes(ctypes.byref(d)) > 0
    bad = _lib.ConvDesc(22, 64, 16, 56, 56, 144, 16, 57, 56, 1, 3, 3, 1, 1, 1, 0, 1, 1)   # Ho inconsistent
    assert lib.zsv_conv3d_wgrad_workspace_bytes(ctypes.byref(bad)) == 0
    assert lib.zsv_conv3d_fwd(ctypes.byref(bad), None, None, None, None, 0, None, 0, None) == 1      # ZSV_E_BAD_SHAPE
    assert lib.zsv_conv3d_fwd(ctypes.byref(d), None, None, None, None, 0, None, 0, None) == 2        # ZSV_E_NULL
    assert lib.zsv_bn_workspace_bytes(22, 144, 50176) > 0
    assert lib.zsv_bn_workspace_bytes(0, 144, 50176) == 0


def test_no_cpu_fallback():
    x = torch.randn(1, 3, 2, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv3d(x, torch.randn(4, 3, 1, 3, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.relu(x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layers.BatchNorm3d(3)(x)
    model = network.get_network(make_opt("r2plus1d_18"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(1, 1, 3, 4, 16, 16))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "zeroshotvideoclassification_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{fn} imports oracle/"


@pytest.mark.parametrize("name", ["r2plus1d_18", "r3d_18", "c3d"])
def test_state_dict_keys_match_the_reference(name):
    golden = {"r2plus1d_18": "r2plus1d_A", "r3d_18": "r3d_small", "c3d": "c3d_eval"}[name]
    g = load_golden(golden)
    model = network.get_network(make_opt(name))
    params = [k for k, _ in model.named_parameters()]
    expected = [str(k) for k in g["live_params"]] + [str(k) for k in g["dead_params"]]
    assert sorted(params) == sorted(expected)
    if name == "r2plus1d_18":
        sd = model.state_dict()
        assert len(sd) == 304
        assert sum(p.numel() for p in model.parameters()) == 36792537
        live = sum(dict(model.named_parameters())[str(k)].numel() for k in g["live_params"])
        assert live == 31716681
        for key in ["model.stem.0.weight", "model.stem.4.running_var", "model.layer1.0.conv1.0.3.weight",
                    "model.layer2.0.downsample.1.num_batches_tracked", "model.fc.bias",
                    "output2emb_proj.layers.1.weight", "encoder.layers.5.norm2.bias", "t_pos_embeds.weight"]:
            assert key in sd, key
        assert tuple(sd["model.layer2.0.conv1.0.0.weight"].shape) == (230, 64, 1, 3, 3)   # midplanes rule
        assert tuple(sd["model.layer4.0.conv2.0.0.weight"].shape) == (921, 512, 1, 3, 3)


def test_constructor_signatures_match_the_reference_surface():
    def params(fn):
        return list(inspect.signature(fn).parameters)
    assert params(network.get_network) == ["opt"]
    assert params(network.Model.__init__) == ["self", "network", "fixconvs", "nopretrained"]
    assert inspect.signature(network.Model.__init__).parameters["nopretrained"].default is False
    assert params(network.C3D.__init__) == ["self", "fixconvs", "nopretrained"]
    assert inspect.signature(network.C3D.__init__).parameters["nopretrained"].default is True
    assert params(network.MLP.__init__) == ["self", "input_dim", "hidden_dim", "output_dim", "num_layers", "last_activate"]
    assert params(network.ResNet18.__init__) == ["self", "network", "fixconvs", "nopretrained"]
    for fac in (resnet.r3d_18, resnet.mc3_18, resnet.r2plus1d_18):
        assert params(fac) == ["pretrained", "progress", "kwargs"]
    assert params(resnet.VideoResNet.__init__) == ["self", "block", "conv_makers", "layers", "stem", "num_classes",
                                                   "zero_init_residual"]
    for cls in (resnet.Conv3DSimple, resnet.Conv2Plus1D, resnet.Conv3DNoTemporal):
        assert params(cls.__init__) == ["self", "in_planes", "out_planes", "midplanes", "stride", "padding"]
    assert resnet.Conv3DNoTemporal.get_downsample_stride(2) == (1, 2, 2)
    assert resnet.Conv2Plus1D.get_downsample_stride(2) == (2, 2, 2)
    assert params(resnet.BasicBlock.__init__) == ["self", "inplanes", "planes", "conv_builder", "stride", "downsample"]
    assert resnet.BasicBlock.expansion == 1 and resnet.Bottleneck.expansion == 4
    assert resnet.__all__ == ["r3d_18", "mc3_18", "r2plus1d_18"]


def test_get_network_dispatch_and_errors():
    assert isinstance(network.get_network(make_opt("r3d_18")), network.Model)
    assert isinstance(network.get_network(make_opt("c3d")), network.C3D)
    with pytest.raises(Exception, match="not available"):
        network.get_network(make_opt("resnet50"))
    with pytest.raises(RuntimeError):
        resnet.r2plus1d_18(pretrained=True)
    frozen = network.get_network(make_opt("r2plus1d_18", fixconvs=True))
    assert not any(p.requires_grad for p in frozen.model.parameters())
    assert all(p.requires_grad for p in frozen.output2emb_proj.parameters())


def test_checkpoint_round_trip_with_module_prefix(tmp_path):
    """main.py:114-124,361-365: checkpoints carry a 'module.' prefix and are key-intersected."""
    model = network.get_network(make_opt("r2plus1d_18"))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=3, bn_jitter=True))
    ckpt = {"state_dict": {"module." + k: v for k, v in model.state_dict().items()}, "opt": {}, "accuracy": 1.0}
    path = tmp_path / "checkpoint.pth.tar"
    torch.save(ckpt, path)
    other = network.get_network(make_opt("r2plus1d_18"))
    j = len("module.")
    weights = torch.load(path)["state_dict"]
    model_dict = other.state_dict()
    weights = {k[j:]: v for k, v in weights.items() if k[j:] in model_dict.keys()}
    model_dict.update(weights)
    other.load_state_dict(model_dict)
    for (k, a), (_, b) in zip(model.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k


def test_save_checkpoint_and_load_weights_helpers(tmp_path):
    from zeroshotvideoclassification_amd import train
    model = network.get_network(make_opt("r3d_18"))
    model.load_state_dict(synthetic.keyed_state_dict(model.state_dict(), seed=5))
    path = str(tmp_path / "checkpoint.pth.tar")
    train.save_checkpoint(model, path, opt={"network": "r3d_18"}, accuracy=12.5)
    ckpt = torch.load(path, weights_only=False)
    assert set(ckpt) == {"state_dict", "opt", "accuracy"} and ckpt["accuracy"] == 12.5
    assert all(k.startswith("module.") for k in ckpt["state_dict"])
    ckpt["state_dict"]["module.not_in_model"] = torch.zeros(1)            # dropped by the key intersection
    torch.save(ckpt, path)
    other = network.get_network(make_opt("r3d_18"))
    assert train.load_weights(other, path) == len(model.state_dict())
    for (k, a), (_, b) in zip(model.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k


def test_synthetic_inputs_follow_the_input_contract():
    x = synthetic.synthetic_clips(2, 4, 16)
    assert x.shape == (2, 1, 3, 4, 16, 16) and x.dtype == torch.float32
    assert x.min() >= -0.5 and x.max() <= 0.0                     # (u8/255 - 1)/2, transforms.py:116-117
    assert torch.equal(x, synthetic.synthetic_clips(2, 4, 16))
    assert not torch.equal(x, synthetic.synthetic_clips(2, 4, 16, rank=1))
    labels, z = synthetic.synthetic_targets(5, 400)
    assert z.shape == (5, 300) and torch.allclose(z.norm(dim=1), torch.ones(5), atol=1e-6)
    sd = network.get_network(make_opt("r2plus1d_18")).state_dict()
    a = synthetic.keyed_state_dict(sd, seed=0)
    b = synthetic.keyed_state_dict(sd, seed=0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    w = a["model.layer1.0.conv1.0.0.weight"]
    assert abs(w.std().item() / (2.0 / (144 * 9)) ** 0.5 - 1) < 0.02     # kaiming fan_out (resnet.py:228)


def test_bench_refuses_to_run_without_the_gpu():
    """bench.py measures the HIP path only: on a host without an MI355X it must exit loudly, not fall back."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert r.returncode != 0
    assert "MI355X" in (r.stderr + r.stdout)
    assert "clips/s" not in r.stdout


def test_bench_self_launch_builds_one_child_per_gpu():
    """`python bench.py --gpus N` without a launcher (the driver's SCALE run): one child per GPU with the
    environment torch.distributed.run would give it, rendezvous on 127.0.0.1, the user's flags passed through;
    fewer visible devices than N is a loud non-zero exit; the N = 1 path never spawns."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    jobs = bench.child_commands(4, argv, 29555)
    assert len(jobs) == 4
    for rank, (cmd, env) in enumerate(jobs):
        assert cmd[0] == sys.executable and cmd[1] == os.path.join(ROOT, "bench.py") and cmd[2:] == argv
        assert env["RANK"] == env["LOCAL_RANK"] == str(rank) and env["WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29555"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    if torch.cuda.device_count() >= 2:
        pytest.skip("enough GPUs are present: the launch itself is exercised by bench.py --gpus 2")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "HIP device(s) are visible" in r.stderr and "clips/s" not in r.stdout
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src                # children are started, never exec'ed into


def test_environment_switches_are_snapshotted_and_reloaded():
    """The library reads its ZSV_* switches once (csrc/knobs.h), not with getenv() on every launch; a process that flips one
    calls _lib.reload_knobs(), which re-reads them (and tells the weight-panel cache) only when something changed.  No audit
    hook is installed unless ZSV_WATCH_ENV asks for it."""
    from ctypes import byref
    from zeroshotvideoclassification_amd import _lib, ops
    d = ops.conv_desc((2, 64, 8, 56, 56), (144, 64, 1, 3, 3), 1, (0, 1, 1))
    saved = os.environ.pop("ZSV_NO_WINO", None)
    try:
        lib = _lib.load()
        _lib.reload_knobs()
        base = lib.zsv_conv3d_fwd_workspace_bytes(byref(d))
        gen = _lib.knob_generation()
        os.environ["ZSV_NO_WINO"] = "1"
        assert lib.zsv_conv3d_fwd_workspace_bytes(byref(d)) == base          # the snapshot, not the environment, decides
        assert _lib.load().zsv_conv3d_fwd_workspace_bytes(byref(d)) == base  # ... and load() alone does not look either
        assert _lib.reload_knobs() is True
        assert _lib.knob_generation() == gen + 1
        direct = lib.zsv_conv3d_fwd_workspace_bytes(byref(d))
        assert direct != base                               # the Winograd-form kernel and the direct kernel pack weights differently
        assert _lib.reload_knobs() is False                 # nothing changed: cached panels stay valid
        assert _lib.knob_generation() == gen + 1
        del os.environ["ZSV_NO_WINO"]
        _lib.reload_knobs()
        assert lib.zsv_conv3d_fwd_workspace_bytes(byref(d)) == base
        os.environ["UNRELATED_VARIABLE"] = "1"
        assert _lib.reload_knobs() is False
        del os.environ["UNRELATED_VARIABLE"]
    finally:
        os.environ.pop("ZSV_NO_WINO", None)
        if saved is not None:
            os.environ["ZSV_NO_WINO"] = saved
        _lib.reload_knobs()
    lib_src = open(os.path.join(ROOT, "zeroshotvideoclassification_amd", "_lib.py")).read()
    assert lib_src.count("sys.addaudithook(") == 1 and 'os.environ.get("ZSV_WATCH_ENV")' in lib_src   # opt-in only
    csrc = os.path.join(ROOT, "zeroshotvideoclassification_amd", "csrc")
    src = "".join(open(os.path.join(csrc, f)).read() for f in os.listdir(csrc) if f.endswith(".hip") and f != "knobs.hip")
    assert "getenv(" not in src                         # the launch path never walks the environment
