#!/usr/bin/env python
"""conv1x1_pipe's wave layout per shape: waves along the pixel blocks (g1_wr: 1 / 2 / 4) x pixel blocks per wave (g1_mbw: 2 / 4 / 8), the
shapes of tools/ab_conv1x1_shapes.py, HIP-event timing on rotating inputs ("dbg" 51: without conv1x1_deepk).  0/0 = the launcher's own
choice.      gpurun -- 'python tools/g1_sweep.py'"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402
from tools.ab_conv1x1_shapes import SHAPES  # noqa: E402

DEV = "cuda:0"
COMBOS = [(0, 0, 0), (1, 4, 0), (1, 8, 0), (1, 4, 2), (1, 8, 2), (2, 4, 0), (2, 4, 2), (2, 2, 2), (4, 4, 0)]
WARM, ITERS = 2000, 200      # WARM: ~50-100 ms of launches first -- the clocks ramp; a cold burst reads 2x the in-step time


def main():
    only = sys.argv[1:]
    ops.set_option("dbg", 51)
    print("%-28s %5s %5s %7s | " % ("shape", "Cin", "Cout", "pixels") + " ".join("%d/%d/%d" % c for c in COMBOS) + "   (wr / mbw / ntw, 0 = auto)")
    for name, n, h, w, cin, cout, add in SHAPES:
        if only and not any(o in name for o in only):
            continue
        rs = np.random.RandomState(0)
        in_bytes = n * h * w * cin * 2
        nrot = max(2, int(np.ceil(1.3 * (256 << 20) / in_bytes)))
        base = torch.from_numpy(np.maximum(rs.randn(n, h, w, cin), 0).astype(np.float16)).to(DEV)
        xs = [base.clone() for _ in range(nrot)]
        pk = ops.pack_conv_weights(torch.from_numpy((rs.randn(1, 1, cin, cout) * 0.05).astype(np.float32)).to(DEV), torch.float16)
        b = torch.zeros(cout, dtype=torch.float32, device=DEV)
        y = torch.zeros((n, h, w, cout), dtype=torch.float16, device=DEV)
        row = []
        for wr, mbw, ntw in COMBOS:
            ops.set_option("g1_wr", wr)
            ops.set_option("g1_mbw", mbw)
            ops.set_option("g1_ntw", ntw)
            for i in range(WARM):
                ops.conv2d_nhwc(xs[i % nrot], pk, b, 1, "SAME", True, out=y, accumulate=bool(add))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(ITERS):
                ops.conv2d_nhwc(xs[i % nrot], pk, b, 1, "SAME", True, out=y, accumulate=bool(add))
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / ITERS * 1e3)
        print("%-28s %5d %5d %7d | " % (name + (" add" if add and "add" not in name else ""), cin, cout, n * h * w) + " ".join("%6.1f" % r for r in row))
    for k in ("g1_wr", "g1_mbw", "g1_ntw", "dbg"):
        ops.set_option(k, 0)


if __name__ == "__main__":
    main()
