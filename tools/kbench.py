#!/usr/bin/env python
"""Per-layer microbenchmark: every conv/pool launch of SqueezeDet (batch 32, 375x1242, fp16 by
default) timed in isolation with HIP events, specialised kernels vs the generic ones (A/B in one
process, interleaved).  Prints ms, algorithmic GB/s and TFLOP/s per layer.

    python tools/kbench.py [--batch 32] [--dtype fp16] [--iters 20] [--only expand1x1]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402

FIRES = [("fire2", 16, 64, 64), ("fire3", 16, 64, 64), ("fire4", 32, 128, 128), ("fire5", 32, 128, 128),
         ("fire6", 48, 192, 192), ("fire7", 48, 192, 192), ("fire8", 64, 256, 256), ("fire9", 64, 256, 256),
         ("fire10", 96, 384, 384), ("fire11", 96, 384, 384)]


def layers(h, w):
    out = [("conv1", "conv", 3, 64, 3, 2, h, w)]
    o = lambda n: -(-n // 2)
    h, w = o(h), o(w)
    out.append(("pool1", "pool", 64, 64, 3, 2, h, w))
    h, w = o(h), o(w)
    c = 64
    for name, s, e1, e3 in FIRES:
        out.append((name + "/squeeze1x1", "conv", c, s, 1, 1, h, w))
        out.append((name + "/expand1x1", "conv", s, e1, 1, 1, h, w))
        out.append((name + "/expand3x3", "conv", s, e3, 3, 1, h, w))
        c = e1 + e3
        if name in ("fire3", "fire5"):
            out.append(("pool" + name[4:], "pool", c, c, 3, 2, h, w))
            h, w = o(h), o(w)
    out.append(("conv12", "conv", c, 72, 3, 1, h, w))
    return out


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        st.record()
        for _ in range(iters):
            fn()
        en.record()
        en.synchronize()
        best = min(best, st.elapsed_time(en) / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--opt", action="append", default=[], help="tuning knob name=value (sqdet_set_option), repeatable")
    args = ap.parse_args()
    for kv in args.opt:
        k, v = kv.split("=")
        ops.set_option(k, int(v))
    if args.opt:
        print("options:", " ".join(args.opt))
    dt = torch.float16 if args.dtype == "fp16" else torch.float32
    esz = 2 if args.dtype == "fp16" else 4
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(0)
    print("%-22s %9s %9s %8s | %9s %9s %8s" % ("layer", "auto_ms", "GB/s", "TF/s", "generic", "GB/s", "TF/s"))
    tot = [0.0, 0.0]
    for name, kind, cin, cout, k, s, h, w in layers(args.height, args.width):
        if args.only and args.only not in name:
            continue
        x = torch.randn((args.batch, h, w, cin), generator=g).to(dev, dt)
        ho, wo = -(-h // s), -(-w // s)
        if kind == "conv":
            wt = (torch.randn((k, k, cin, cout), generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(dev)
            b = torch.zeros(cout, device=dev)
            pk = ops.pack_conv_weights(wt, dt)
            y = torch.empty((args.batch, ho, wo, cout), dtype=dt, device=dev)
            fn = lambda: ops.conv2d_nhwc(x, pk, b, s, "SAME", True, out=y)
            flops = 2.0 * k * k * cin * cout * args.batch * ho * wo
            nbytes = (x.numel() + y.numel() + wt.numel()) * esz
        else:
            fn = lambda: ops.maxpool_nhwc(x, k, s, "SAME")
            flops = 0.0
            nbytes = (x.numel() + args.batch * ho * wo * cin) * esz
        res = []
        for algo in (0, 1):
            ops.set_option("conv_algo", algo)
            res.append(timeit(fn, args.iters))
        ops.set_option("conv_algo", 0)
        tot[0] += res[0]
        tot[1] += res[1]
        f = lambda ms: (nbytes / ms / 1e6, flops / ms / 1e9)
        print("%-22s %9.4f %9.1f %8.2f | %9.4f %9.1f %8.2f" % ((name, res[0]) + f(res[0]) + (res[1],) + f(res[1])))
        del x
    # fused stem
    if not args.only or "stem" in args.only:
        x = torch.randn((args.batch, args.height, args.width, 3), generator=g).to(dev, dt)
        wt = (torch.randn((3, 3, 3, 64), generator=g) * 0.27).to(dev)
        b = torch.zeros(64, device=dev)
        pk = ops.pack_conv_weights(wt, dt)
        ms = timeit(lambda: ops.stem_conv_pool(x, pk, b, "SAME", "SAME"), args.iters)
        hp, wp = -(-(-(-args.height // 2)) // 2), -(-(-(-args.width // 2)) // 2)
        nbytes = (x.numel() + args.batch * hp * wp * 64) * esz
        print("%-22s %9.4f %9.1f   (fused conv1+pool1)" % ("stem", ms, nbytes / ms / 1e6))
    print("sum auto %.4f ms   sum generic %.4f ms" % tuple(tot))
    # whole fire modules: one fused launch vs squeeze -> expand1x1 / expand3x3
    if not args.only or "fire" in args.only:
        print("%-10s %10s %10s   (fused vs three launches, ms)" % ("module", "fused", "separate"))
        o = lambda n: -(-n // 2)
        h, w = o(o(args.height)), o(o(args.width))
        c = 64
        ft = st = 0.0
        for name, s, e1, e3 in FIRES:
            x = torch.randn((args.batch, h, w, c), generator=g).clamp_(min=0).to(dev, dt)
            mk = lambda k, ci, co: ops.pack_conv_weights((torch.randn((k, k, ci, co), generator=g) * (2.0 / (k * k * ci)) ** 0.5).to(dev), dt)
            ps, p1, p3 = mk(1, c, s), mk(1, s, e1), mk(3, s, e3)
            bz = [torch.zeros(n_, device=dev) for n_ in (s, e1, e3)]
            fn = lambda: ops.fire(x, ps, bz[0], p1, bz[1], p3, bz[2])
            ops.set_option("fire_fuse", 1)
            t1 = timeit(fn, args.iters)
            ops.set_option("fire_fuse", 2)
            t2 = timeit(fn, args.iters)
            ops.set_option("fire_fuse", 0)
            ft += t1
            st += t2
            print("%-10s %10.4f %10.4f" % (name, t1, t2))
            c = e1 + e3
            if name in ("fire3", "fire5"):
                h, w = o(h), o(w)
            del x
        print("%-10s %10.4f %10.4f" % ("sum", ft, st))


if __name__ == "__main__":
    main()
