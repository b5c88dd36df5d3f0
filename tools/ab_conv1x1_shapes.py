#!/usr/bin/env python
"""Per-shape A/B of the two deep-K 1x1 kernels -- conv1x1_deepk (weights resident in LDS, activations streamed; conv1x1k.hip) and
conv1x1_tile (workgroup GEMM tile; gemm1x1.hip, "dbg" 51) -- on the plain 1x1 shapes of ResNet50 res3 / res4 at the training batch
(8 x 375x1242) and of SqueezeDet+; the backward-data convs of the same layers have these shapes with Cin / Cout swapped.  HIP-event
timing, inputs rotating over > 333 MB.
    gpurun -- 'python tools/ab_conv1x1_shapes.py'"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [("res4 2a fwd / 2c dgrad", 8, 24, 78, 1024, 256), ("res4 2c fwd / 2a dgrad", 8, 24, 78, 256, 1024),
          ("res3 2a fwd / 2c dgrad", 8, 47, 156, 512, 128), ("res4 branch1-like", 8, 24, 78, 512, 1024),
          ("plus fire9 squeeze", 8, 22, 76, 512, 384), ("plus fire9 e1 / sq dgrad", 8, 22, 76, 384, 256),
          ("plus fire6 squeeze", 8, 45, 153, 256, 288), ("plus fire6 e1", 8, 45, 153, 288, 192)]
WARM, ITERS = 3, 30


def time_shape(n, h, w, cin, cout):
    rs = np.random.RandomState(0)
    in_bytes = n * h * w * cin * 2
    nrot = max(2, int(np.ceil(1.3 * (256 << 20) / in_bytes)))
    base = torch.from_numpy(np.maximum(rs.randn(n, h, w, cin), 0).astype(np.float16)).to(DEV)
    xs = [base.clone() for _ in range(nrot)]
    pk = ops.pack_conv_weights(torch.from_numpy((rs.randn(1, 1, cin, cout) * 0.05).astype(np.float32)).to(DEV), torch.float16)
    b = torch.zeros(cout, dtype=torch.float32, device=DEV)
    y = torch.empty((n, h, w, cout), dtype=torch.float16, device=DEV)
    res = []
    for dbg in (0, 51):
        ops.set_option("dbg", dbg)
        for i in range(WARM):
            ops.conv2d_nhwc(xs[i % nrot], pk, b, 1, "SAME", True, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(ITERS):
            ops.conv2d_nhwc(xs[i % nrot], pk, b, 1, "SAME", True, out=y)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / ITERS * 1e3)
        if dbg == 0:
            ref = y.clone()
        else:
            same = bool(torch.equal(ref, y))
    ops.set_option("dbg", 0)
    return res[0], res[1], same, in_bytes + n * h * w * cout * 2


def main():
    print("%-28s %6s %6s %8s | %10s %10s %6s | %s" % ("shape", "Cin", "Cout", "pixels", "deepk us", "tile us", "ratio", "GB/s deepk, bitwise"))
    for name, n, h, w, cin, cout in SHAPES:
        a, b, same, nbytes = time_shape(n, h, w, cin, cout)
        print("%-28s %6d %6d %8d | %10.1f %10.1f %6.2f | %.0f %s" % (name, cin, cout, n * h * w, a, b, a / b, nbytes / a / 1e3, "same" if same else "DIFFERENT"))


if __name__ == "__main__":
    main()
