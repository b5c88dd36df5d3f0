#!/usr/bin/env python
"""Per-shape A/B of the deep-K 1x1 kernels -- conv1x1_deepk (weights resident in LDS, activations streamed; conv1x1k.hip; default where
eligible), conv1x1_pipe (the pipelined workgroup GEMM tile of round 6; gemm1x1.hip, "dbg" 51 = without deepk) and conv1x1_tile (its
round-2..5 predecessor, "dbg" 57) -- on the 1x1 shapes of ResNet50 res2..res5 at the training batch (8 x 375x1242) and of SqueezeDet+ at
batch 8; the backward-data convs of the same layers have these shapes with Cin / Cout swapped.  `add` = the residual-accumulate form
(y += conv, ReLU: ResNet50's branch2c).  HIP-event timing, inputs rotating over > 333 MB.
    gpurun -- 'python tools/ab_conv1x1_shapes.py'"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [("res4 2a fwd / 2c dgrad", 8, 24, 78, 1024, 256, 0), ("res4 2c fwd / 2a dgrad", 8, 24, 78, 256, 1024, 0),
          ("res4 2c fwd add", 8, 24, 78, 256, 1024, 1),
          ("res3 2a fwd / 2c dgrad", 8, 47, 156, 512, 128, 0), ("res3 2c fwd add", 8, 47, 156, 128, 512, 1),
          ("res2 2a fwd", 8, 94, 311, 256, 64, 0), ("res2 2c fwd add", 8, 94, 311, 64, 256, 1),
          ("res5 2a fwd", 8, 24, 78, 2048, 512, 0), ("res5 2c fwd add", 8, 24, 78, 512, 2048, 1),
          ("res4 branch1-like", 8, 24, 78, 512, 1024, 0),
          ("plus fire9 squeeze", 8, 22, 76, 512, 384, 0), ("plus fire9 e1 / sq dgrad", 8, 22, 76, 384, 256, 0),
          ("plus fire6 squeeze", 8, 45, 153, 256, 288, 0), ("plus fire6 e1", 8, 45, 153, 288, 192, 0),
          ("plus fire2 squeeze", 8, 92, 309, 96, 96, 0), ("plus fire3 e1", 8, 92, 309, 96, 128, 0),
          ("sqdet fire6 squeeze b32", 32, 24, 78, 256, 48, 0), ("sqdet fire11 squeeze b32", 32, 24, 78, 768, 96, 0)]
WARM, ITERS = 2000, 200      # WARM: ~50-100 ms of launches first -- the clocks ramp; a cold burst reads 2x the in-step time


def time_shape(n, h, w, cin, cout, add):
    rs = np.random.RandomState(0)
    in_bytes = n * h * w * cin * 2
    nrot = max(2, int(np.ceil(1.3 * (256 << 20) / in_bytes)))
    base = torch.from_numpy(np.maximum(rs.randn(n, h, w, cin), 0).astype(np.float16)).to(DEV)
    xs = [base.clone() for _ in range(nrot)]
    pk = ops.pack_conv_weights(torch.from_numpy((rs.randn(1, 1, cin, cout) * 0.05).astype(np.float32)).to(DEV), torch.float16)
    b = torch.zeros(cout, dtype=torch.float32, device=DEV)
    y0 = torch.from_numpy(rs.randn(n, h, w, cout).astype(np.float16)).to(DEV)
    y = torch.empty((n, h, w, cout), dtype=torch.float16, device=DEV)
    res, ref, same = [], None, True
    for dbg in (0, 51, 57):
        ops.set_option("dbg", dbg)

        def run(i):
            ops.conv2d_nhwc(xs[i % nrot], pk, b, 1, "SAME", True, out=y, accumulate=bool(add))
        for i in range(WARM):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(ITERS):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / ITERS * 1e3)
        # bitwise check on a fresh output (the accumulate form adds into y)
        y.copy_(y0)
        run(0)
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        else:
            same = same and bool(torch.equal(ref, y))
    ops.set_option("dbg", 0)
    return res, same, in_bytes + (2 if add else 1) * n * h * w * cout * 2 + cin * cout * 2, 2.0 * n * h * w * cin * cout


def main():
    print("%-28s %5s %5s %7s | %8s %8s %8s | %6s | %s" % ("shape", "Cin", "Cout", "pixels", "default", "pipe us", "tile us", "tile/pipe",
                                                          "default: GB/s, TF/s, bitwise"))
    for name, n, h, w, cin, cout, add in SHAPES:
        r, same, nbytes, flops = time_shape(n, h, w, cin, cout, add)
        print("%-28s %5d %5d %7d | %8.1f %8.1f %8.1f | %6.2f | %.0f %.0f %s" % (name, cin, cout, n * h * w, r[0], r[1], r[2], r[2] / r[1],
                                                                            nbytes / r[0] / 1e3, flops / r[0] / 1e6, "same" if same else "DIFFERENT"))


if __name__ == "__main__":
    main()
