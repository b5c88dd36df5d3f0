#!/usr/bin/env python
"""conv12 of SqueezeDet (768 -> 72, 24x78) timed at batch 32 / 20 / 8 / 1 with rotating inputs, and a checksum of the
output for bitwise comparison between two builds:   python tools/convdet_ab.py   (CD_DTYPE=fp32 for float32)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402
from tools.kbench import timeit  # noqa: E402

dev = "cuda:0"
rs = np.random.RandomState(0)
w = torch.from_numpy((rs.randn(3, 3, 768, 72) * 0.02).astype(np.float32)).to(dev)
DT = torch.float32 if os.environ.get("CD_DTYPE") == "fp32" else torch.float16
pw = ops.pack_conv_weights(w, DT)
b = torch.from_numpy(rs.randn(72).astype(np.float32)).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
for batch in (32, 20, 8, 1):
    xs = [torch.randn(batch, 24, 78, 768, device=dev, generator=g).to(DT) for _ in range(4)]
    y = ops.conv2d_nhwc(xs[0], pw, b, relu=False)
    chk = int(y.view(torch.int16 if DT == torch.float16 else torch.int32).to(torch.int64).sum().item())
    k = [0]

    def fn():
        k[0] = (k[0] + 1) & 3
        ops.conv2d_nhwc(xs[k[0]], pw, b, relu=False)
    us = [timeit(fn, 50) * 1e3 for _ in range(3)]
    print("batch %2d: %s us   checksum %d" % (batch, " ".join("%6.1f" % v for v in us), chk), flush=True)
