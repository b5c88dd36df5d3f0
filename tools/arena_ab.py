#!/usr/bin/env python
"""Does it matter WHERE a tensor lives?  The stand-alone 1x1 convs of tools/fire1x1_standalone.py on (a) tensors of their own (one
allocation each, as torch hands them out) and (b) views into ONE 3 GiB allocation (inputs / outputs at 256 MiB-aligned offsets), both
rotating over more than the 256 MiB Infinity Cache; N launches captured in one hipGraph (no host time between them), best of 3 replays.
    python tools/arena_ab.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402

SHAPES = [("fire2/squeeze1x1", 94, 311, 64, 16), ("fire2/expand1x1", 94, 311, 16, 64), ("fire4/expand1x1", 47, 156, 32, 128),
          ("fire5/squeeze1x1", 47, 156, 256, 32), ("fire11/squeeze1x1", 24, 78, 768, 96)]
B, NROT = 32, 6


def graph_time(fns, reps=3):
    for f in fns:
        f()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for f in fns:
                f()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        st.record()
        g.replay()
        en.record()
        en.synchronize()
        best = min(best, st.elapsed_time(en) / len(fns))
    return best * 1e3


def main():
    dev = "cuda:0"
    rs = np.random.RandomState(0)
    arena = torch.empty(3 << 30, dtype=torch.uint8, device=dev)
    print("arena base %% 1 GiB = %d MiB" % ((arena.data_ptr() % (1 << 30)) >> 20))
    print("%-20s %10s %12s %12s" % ("layer", "alg MB", "own us (GB/s)", "arena us (GB/s)"))
    for name, h, w, cin, cout in SHAPES:
        x0 = torch.from_numpy(np.maximum(rs.randn(B, h, w, cin), 0).astype(np.float16)).to(dev)
        wk = torch.from_numpy((rs.randn(1, 1, cin, cout) * 0.1).astype(np.float32)).to(dev)
        b = torch.zeros(cout, dtype=torch.float32, device=dev)
        pk = ops.pack_conv_weights(wk, torch.float16)
        alg = (B * h * w * (cin + cout) + cin * cout) * 2 + 4 * cout
        # (a) tensors of their own
        xs = [x0.clone() for _ in range(NROT)]
        ys = [torch.empty((B, h, w, cout), dtype=torch.float16, device=dev) for _ in range(NROT)]
        t_own = graph_time([(lambda i=i: ops.conv2d_nhwc(xs[i], pk, b, 1, "SAME", True, out=ys[i])) for i in range(NROT)] * 3)
        ref = ys[0].clone()
        del xs, ys
        # (b) views into the arena: slot k at k * 256 MiB
        def view(k, c):
            n = B * h * w * c
            return arena[k * (256 << 20): k * (256 << 20) + n * 2].view(torch.float16).view(B, h, w, c)
        xa = [view(k, cin) for k in range(NROT)]
        ya = [view(NROT + k, cout) for k in range(NROT)]
        for t in xa:
            t.copy_(x0)
        t_ar = graph_time([(lambda i=i: ops.conv2d_nhwc(xa[i], pk, b, 1, "SAME", True, out=ya[i])) for i in range(NROT)] * 3)
        assert torch.equal(ref, ya[0])
        print("%-20s %10.1f %7.2f (%5.0f) %7.2f (%5.0f)" % (name, alg / 1e6, t_own, alg / t_own / 1e3, t_ar, alg / t_ar / 1e3))
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
