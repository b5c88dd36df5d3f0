#!/usr/bin/env python
"""Late-map fire modules (fire6..fire11 at batch 32, 24x78, float16): the chain kernel (expand + next squeeze in one
launch, sqdet_fire_chain_fwd) against the one-launch fused fire module (sqdet_fire_fwd), HIP events, interleaved.

    python tools/chainbench.py [--batch 32] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402
from tools.kbench import timeit as timeit_eager  # noqa: E402


def timeit(fn, iters):
    """iters launches captured in ONE hipGraph (no host issue time between them), best of 3 replays."""
    if os.environ.get("CHAINBENCH_EAGER") == "1":
        return timeit_eager(fn, iters)
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        st.record()
        g.replay()
        en.record()
        en.synchronize()
        best = min(best, st.elapsed_time(en) / iters)
    return best

LATE = [("fire6", 256, 48, 192), ("fire7", 384, 48, 192), ("fire8", 384, 64, 256), ("fire9", 512, 64, 256),
        ("fire10", 512, 96, 384), ("fire11", 768, 96, 384)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--h", type=int, default=24)
    ap.add_argument("--w", type=int, default=78)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dbg", default="", help="comma list of dbg knob values to time the chain kernel under (wrong results)")
    args = ap.parse_args()
    dev, dt = "cuda:0", torch.float16
    g = torch.Generator(device="cpu").manual_seed(0)
    mkw = lambda k, ci, co: (torch.randn((k, k, ci, co), generator=g) * (2.0 / (k * k * ci)) ** 0.5).to(dev)
    print("%-8s %10s %10s %10s %10s %9s" % ("module", "fused_ms", "squeeze_ms", "chain_ms", "chain+y", "chain TF/s"))
    tot = [0.0, 0.0, 0.0]
    npx = args.batch * args.h * args.w
    for i, (name, cin, s, e) in enumerate(LATE):
        s2 = LATE[i + 1][2] if i + 1 < len(LATE) else 0
        x = torch.randn((args.batch, args.h, args.w, cin), generator=g).clamp_(min=0).to(dev, dt)
        ws, w1, w3 = mkw(1, cin, s), mkw(1, s, e), mkw(3, s, e)
        wn = mkw(1, 2 * e, s2) if s2 else None
        ps, p1, p3 = [ops.pack_conv_weights(w_, dt) for w_ in (ws, w1, w3)]
        bz = [torch.zeros(n_, device=dev) for n_ in (s, e, e, max(s2, 1))]
        t_fused = timeit(lambda: ops.fire(x, ps, bz[0], p1, bz[1], p3, bz[2]), args.iters)
        sq = torch.empty((args.batch, args.h, args.w, s), dtype=dt, device=dev)
        t_sq = timeit(lambda: ops.conv2d_nhwc(x, ps, bz[0], 1, "SAME", True, out=sq), args.iters)
        chain = ops.FireChainStream(w1, w3, wn, dt)
        t_chain = timeit(lambda: ops.fire_chain(sq, chain, bz[1], bz[2], bz[3] if s2 else None), args.iters)
        t_cy = timeit(lambda: ops.fire_chain(sq, chain, bz[1], bz[2], bz[3] if s2 else None, want_y=True), args.iters)
        flops = 2.0 * npx * (s * e * 10 + 2 * e * s2)
        for d in [int(v) for v in args.dbg.split(",") if v]:
            ops.set_option("dbg", d)
            td = timeit(lambda: ops.fire_chain(sq, chain, bz[1], bz[2], bz[3] if s2 else None), args.iters)
            ops.set_option("dbg", 0)
            print("   dbg %d: %.4f ms" % (d, td))
        print("%-8s %10.4f %10.4f %10.4f %10.4f %9.1f" % (name, t_fused, t_sq, t_chain, t_cy, flops / t_chain / 1e9))
        tot[0] += t_fused
        tot[1] += t_chain
        tot[2] += t_sq if i == 0 else 0.0
    print("sum fused %.4f ms; chain %.4f ms (+ first squeeze %.4f ms)" % tuple(tot))
    # the early maps: fire2 -> fire3's squeeze at 94x311, fire4 -> fire5's squeeze at 47x156
    print("%-8s %10s %10s %10s %10s" % ("module", "fused_ms", "squeeze_ms", "chain_ms", "chain+y"))
    for name, cin, s, e, s2, h, w in (("fire2", 64, 16, 64, 16, 94, 311), ("fire4", 128, 32, 128, 32, 47, 156)):
        x = torch.randn((args.batch, h, w, cin), generator=g).clamp_(min=0).to(dev, dt)
        ws, w1, w3, wn = mkw(1, cin, s), mkw(1, s, e), mkw(3, s, e), mkw(1, 2 * e, s2)
        ps, p1, p3 = [ops.pack_conv_weights(w_, dt) for w_ in (ws, w1, w3)]
        bz = [torch.zeros(n_, device=dev) for n_ in (s, e, e, s2)]
        t_fused = timeit(lambda: ops.fire(x, ps, bz[0], p1, bz[1], p3, bz[2]), args.iters)
        sq = torch.empty((args.batch, h, w, s), dtype=dt, device=dev)
        t_sq = timeit(lambda: ops.conv2d_nhwc(x, ps, bz[0], 1, "SAME", True, out=sq), args.iters)
        chain = ops.FireChainStream(w1, w3, wn, dt)
        t_chain = timeit(lambda: ops.fire_chain(sq, chain, bz[1], bz[2], bz[3]), args.iters)
        t_cy = timeit(lambda: ops.fire_chain(sq, chain, bz[1], bz[2], bz[3], want_y=True), args.iters)
        pn = ops.pack_conv_weights(wn, dt)
        t_one = timeit(lambda: ops.fire_squeeze_next(x, ps, bz[0], p1, bz[1], p3, bz[2], pn, bz[3]), args.iters)
        print("%-8s %10.4f %10.4f %10.4f %10.4f   one launch (x -> next squeeze): %.4f" % (name, t_fused, t_sq, t_chain, t_cy, t_one))
    # the pooled modules from their squeeze tensor: fire3+pool3, fire5+pool5
    print("%-12s %10s %10s" % ("module", "whole_ms", "from_sq_ms"))
    for name, cin, s, e, h, w in (("fire3+pool3", 128, 16, 64, 94, 311), ("fire5+pool5", 256, 32, 128, 47, 156)):
        x = torch.randn((args.batch, h, w, cin), generator=g).clamp_(min=0).to(dev, dt)
        ps, p1, p3 = [ops.pack_conv_weights(w_, dt) for w_ in (mkw(1, cin, s), mkw(1, s, e), mkw(3, s, e))]
        bz = [torch.zeros(n_, device=dev) for n_ in (s, e, e)]
        sq = ops.conv2d_nhwc(x, ps, bz[0], 1, "SAME", True)
        t_whole = timeit(lambda: ops.fire_maxpool(x, ps, bz[0], p1, bz[1], p3, bz[2]), args.iters)
        t_sq = timeit(lambda: ops.fire_expand(sq, p1, bz[1], p3, bz[2], pool=True), args.iters)
        print("%-12s %10.4f %10.4f" % (name, t_whole, t_sq))


if __name__ == "__main__":
    main()
