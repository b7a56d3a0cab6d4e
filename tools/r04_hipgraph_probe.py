"""Can a torch.cuda.CUDAGraph (hipGraph) capture this library's ctypes launches, including work forked to a second stream with
events, torch allocations from the private pool and Tensor.record_stream?  (round 4: the bf16 trunk as two graphs)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zeroshotvideoclassification_amd import amp, inference, ops

dev = torch.device("cuda")
conv = torch.nn.Conv3d(64, 144, (1, 3, 3), padding=(0, 1, 1), bias=False).to(dev)
bn = torch.nn.BatchNorm3d(144).to(dev).train()
u = amp._Unit(conv, bn, True)
x = amp.ncdhw_to_cl_bf16(torch.randn(4, 64, 8, 28, 28, device=dev))
d = u.desc(4, 8, 28, 28)
side = torch.cuda.Stream()


def work():
    blob = inference.pack_conv(d, conv.weight.detach(), None, None)
    z = inference.conv_bf16(d, x, blob, None, False)
    y, mean, invstd, coef = amp.bn_cl_fwd_train(z, bn, None, True, want_coef=True)
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    rec = amp._Record()
    rec.unit, rec.desc, rec.x, rec.clips = u, d, x, None
    with torch.cuda.stream(side):
        dw = amp.Bf16TrainPath._wgrad.__func__(rec, y) if False else None
        t = y.float().sum()                       # torch work on the side stream
    y.record_stream(side)
    main.wait_stream(side)
    return y, t


warm = torch.cuda.Stream()
warm.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(warm):
    for _ in range(3):
        y_ref, t_ref = work()
torch.cuda.current_stream().wait_stream(warm)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y_s, t_s = work()
torch.cuda.synchronize()
x.mul_(0.5)
g.replay()
torch.cuda.synchronize()
y_chk, t_chk = work()
torch.cuda.synchronize()
print("graph replay equals eager:", torch.equal(y_s, y_chk), float(t_s), float(t_chk))
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"replay host cost {1e6 * (t1 - t0) / 50:.1f} us")
