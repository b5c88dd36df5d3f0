#!/usr/bin/env python
"""The fused 7x7 stems stand-alone (batch 8, 375x1242, float16): the default kernel (stem5.hip) against the strip kernel ("stem_algo" 2),
HIP events over 200 launches after a spin-up.     gpurun -- 'python tools/stem_bench.py'"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402

for (n, h, w, cout, cp) in [(8, 375, 1242, 96, "VALID"), (8, 375, 1242, 64, "SAME")]:
    x = torch.randn(n, h, w, 3, device="cuda").half()
    wt = torch.randn(7, 7, 3, cout, device="cuda") * 0.1
    pk = ops.pack_conv_weights(wt, torch.float16)
    b = torch.zeros(cout, device="cuda")
    for algo in (0, 2):
        ops.set_option("stem_algo", algo)
        for i in range(300):
            y = ops.stem_conv_pool(x, pk, b, cp, "VALID")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            y = ops.stem_conv_pool(x, pk, b, cp, "VALID")
        e1.record()
        torch.cuda.synchronize()
        print("7x7 stem, %d couts, %s conv, batch %d: stem_algo %d: %.1f us" % (cout, cp, n, algo, e0.elapsed_time(e1) / 200 * 1e3))
    ops.set_option("stem_algo", 0)
