import os, sys, ctypes, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from zeroshotvideoclassification_amd import _lib, ops
import torch.nn.functional as F
lib = _lib.load()
dev = "cuda"
def run(n, cin, cout, thw, kind):
    t, h, w = thw
    g = torch.Generator().manual_seed(1)
    wt = (torch.randn(cout, cin, 1, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
    x = torch.randn(n, cin, t, h, w, generator=g).to(dev)
    dy = torch.randn(n, cout, t, h, w, generator=g).to(dev)
    d = ops.conv_desc(x.shape, wt.shape, 1, (0, 1, 1))
    res = {}
    for env in ("", "ZSV_WINO_GENERIC_EPILOGUE"):
        if env: os.environ[env] = "1"
        lib2 = _lib.load()
        if kind == "dgrad":
            out = torch.zeros_like(x)
            nb = lib2.zsv_conv3d_dgrad_workspace_bytes(ctypes.byref(d)); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            _lib.check(lib2.zsv_conv3d_dgrad(ctypes.byref(d), dy.data_ptr(), wt.data_ptr(), out.data_ptr(), ws.data_ptr(), nb, None), "dgrad")
        else:
            out = torch.zeros_like(dy)
            nb = lib2.zsv_conv3d_fwd_workspace_bytes(ctypes.byref(d)); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            _lib.check(lib2.zsv_conv3d_fwd(ctypes.byref(d), x.data_ptr(), wt.data_ptr(), None, out.data_ptr(), 0, ws.data_ptr(), nb, None), "fwd")
        torch.cuda.synchronize()
        res[env] = out
        if env: del os.environ[env]
    _lib.load()
    a, b = res[""], res["ZSV_WINO_GENERIC_EPILOGUE"]
    diff = (a - b).abs()
    bad = (diff > 1e-4 * b.abs().max()).nonzero()
    print(kind, n, cin, cout, thw, "max diff", diff.max().item(), "bad", len(bad), "of", a.numel(), "first bad", bad[:3].tolist(), "zero in new", int((a == 0).sum()), "zero in old", int((b == 0).sum()))
    if len(bad):
        chan = torch.unique(bad[:, 1]); print("  bad channels", chan[:20].tolist(), "bad w", torch.unique(bad[:, 4])[:12].tolist(), "bad n", torch.unique(bad[:, 0]).tolist())
run(4, 32, 16, (8, 96, 96), "dgrad")
run(4, 64, 16, (8, 96, 96), "dgrad")
run(4, 16, 32, (8, 96, 96), "fwd")
run(4, 16, 64, (8, 96, 96), "fwd")
run(2, 64, 48, (4, 56, 56), "fwd")
