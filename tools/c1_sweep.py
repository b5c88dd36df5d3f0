#!/usr/bin/env python
"""Stand-alone 1x1 convs of the early fire modules (conv1x1_stream: K <= 4 chunks) under the launcher's knobs -- waves in flight
(`c1_waves`), pixel blocks per wave step (`c1_mt`) -- at batch 32 / 375x1242 / float16, inputs AND outputs rotating over more than the
256 MiB Infinity Cache.  Launches captured in one hipGraph (no host time between them), best of 3 replays.
    python tools/c1_sweep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402
from tools.arena_ab import graph_time  # noqa: E402

SHAPES = [("fire2/squeeze1x1", 94, 311, 64, 16), ("fire2/expand1x1", 94, 311, 16, 64), ("fire3/squeeze1x1", 94, 311, 128, 16),
          ("fire4/squeeze1x1", 47, 156, 128, 32), ("fire4/expand1x1", 47, 156, 32, 128), ("fire6/expand1x1", 24, 78, 48, 192),
          ("fire8/expand1x1", 24, 78, 64, 256), ("fire10/expand1x1", 24, 78, 96, 384)]
WARM, ITERS, B = 3, 20, 32


def main():
    dev = "cuda:0"
    rs = np.random.RandomState(0)
    print("%-20s %9s %6s %9s %8s %8s" % ("layer", "c1_waves", "c1_mt", "us", "GB/s", "of 8TB/s"))
    for name, h, w, cin, cout in SHAPES:
        inb, outb = B * h * w * cin * 2, B * h * w * cout * 2
        nrot = max(2, int(np.ceil(1.3 * (256 << 20) / inb)))
        nrot_y = max(2, int(np.ceil(1.3 * (256 << 20) / outb)))
        x0 = torch.from_numpy(np.maximum(rs.randn(B, h, w, cin), 0).astype(np.float16)).to(dev)
        xs = [x0] + [x0.clone() for _ in range(nrot - 1)]
        ys = [torch.empty((B, h, w, cout), dtype=torch.float16, device=dev) for _ in range(nrot_y)]
        wk = torch.from_numpy((rs.randn(1, 1, cin, cout) * 0.1).astype(np.float32)).to(dev)
        b = torch.zeros(cout, dtype=torch.float32, device=dev)
        pk = ops.pack_conv_weights(wk, torch.float16)
        alg = inb + outb + cin * cout * 2 + 4 * cout
        ref = None
        for waves in (0, 2048, 8192, 16384):
            for mt in (0, 2, 4):
                if waves == 0 and mt != 0 or waves != 0 and mt == 0:
                    continue
                ops.set_option("c1_waves", waves)
                ops.set_option("c1_mt", mt)
                n = max(nrot, nrot_y)
                us = graph_time([(lambda i=i: ops.conv2d_nhwc(xs[i % nrot], pk, b, 1, "SAME", True, out=ys[i % nrot_y])) for i in range(n)] * 2)
                out = ys[(n - 1) % nrot_y]
                if ref is None:
                    ref = out.clone()
                else:
                    assert torch.equal(ref, out), "knobs changed the result"
                print("%-20s %9s %6s %9.2f %8.0f %8.3f" % (name, waves or "default", mt or "auto", us, alg / us / 1e3, alg / us / 1e3 / 8000.0))
        ops.set_option("c1_waves", 0)
        ops.set_option("c1_mt", 0)
        del xs, ys
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
