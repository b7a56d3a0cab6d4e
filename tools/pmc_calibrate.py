"""Streams a known number of bytes through `bn_stats_kernel` (one 16-B/lane read of the tensor, no
writes to speak of) so that a `rocprofv3 --pmc FETCH_SIZE` pass can be calibrated: the S1 mid tensor
22x144x16x56x56 fp32 = 635,830,272 B.  Run under rocprofv3 next to tools/conv_bench.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zeroshotvideoclassification_amd import layers  # noqa: E402

x = torch.randn(22, 144, 16, 56, 56, device="cuda")
bn = layers.BatchNorm3d(144).cuda().train()
with torch.no_grad():
    for _ in range(3):
        bn(x)
torch.cuda.synchronize()
print("ok", x.numel() * 4)
