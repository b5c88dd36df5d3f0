#!/usr/bin/env python
"""conv1x1_pipe's cost per K chunk: the same map and Cout at Cin = 256 .. 2048 (8 .. 64 chunks), HIP events on rotating inputs after a
spin-up; the slope is the K loop's time per 64-byte chunk, the intercept launch + prologue + epilogue.   gpurun -- 'python tools/g1_kslope.py'"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402

DEV = "cuda:0"


def t_conv(n, h, w, cin, cout, iters=300, warm=1500):
    rs = np.random.RandomState(0)
    nrot = 1 if os.environ.get("G1_HOT") else max(2, int(np.ceil(1.3 * (256 << 20) / (n * h * w * cin * 2))))    # G1_HOT: one input, served by the Infinity Cache
    base = torch.from_numpy(np.maximum(rs.randn(n, h, w, cin), 0).astype(np.float16)).to(DEV)
    xs = [base.clone() for _ in range(nrot)]
    pk = ops.pack_conv_weights(torch.from_numpy((rs.randn(1, 1, cin, cout) * 0.05).astype(np.float32)).to(DEV), torch.float16)
    b = torch.zeros(cout, dtype=torch.float32, device=DEV)
    y = torch.empty((n, h, w, cout), dtype=torch.float16, device=DEV)
    for i in range(warm):
        ops.conv2d_nhwc(xs[i % nrot], pk, b, 1, "SAME", True, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        ops.conv2d_nhwc(xs[i % nrot], pk, b, 1, "SAME", True, out=y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ops.set_option("dbg", int(os.environ.get("G1_DBG", "51")))
    for (n, h, w, cout) in [(8, 24, 78, 256), (1, 8, 8, 256), (8, 24, 78, 1024), (8, 47, 156, 128)]:
        ts = []
        for cin in (256, 512, 1024, 2048):
            ts.append(t_conv(n, h, w, cin, cout))
        slope = (ts[3] - ts[1]) / (64 - 16)
        print("%6d px, Cout %4d: Cin 256 / 512 / 1024 / 2048 -> %s us; per chunk %.3f us (%.0f cycles at 2.3 GHz), intercept %.1f us"
              % (n * h * w, cout, " ".join("%6.1f" % t for t in ts), slope, slope * 2300, ts[1] - 16 * slope))
    ops.set_option("dbg", 0)


if __name__ == "__main__":
    main()
