#!/usr/bin/env python
"""Where a fire_chain launch's time goes, per workgroup (experiment): needs libsqdet_hip.so built with chain.hip compiled
-DSQDET_CHAIN_TIMELINE (SQDET_EXTRA_DEFINES="-DSQDET_CHAIN_TIMELINE" python -m squeezedet_amd.build --force).  For every late
module of SqueezeDet at batch 32 (24 x 78 maps): ITERS launches captured in one hipGraph, event-timed; the timeline of the LAST
launch (eight 100 MHz s_memrealtime stamps per workgroup) relative to the first workgroup's entry:
   entry | loads issued | squeeze tile in LDS | first weight stage | expand1x1 done | expand3x3 done | ring drained | last store issued
    python tools/chain_timeline.py [--batch 32] [--iters 10]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import _lib, ops  # noqa: E402
from tools.chainbench import LATE, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    lib = _lib.lib()
    if not hasattr(lib, "sqdet_debug_chain_timeline"):
        sys.exit("chain.hip was not compiled with -DSQDET_CHAIN_TIMELINE")
    lib.sqdet_debug_chain_timeline.argtypes = [C.c_void_p, C.c_int]
    dev, dt = "cuda:0", torch.float16
    g = torch.Generator(device="cpu").manual_seed(0)
    mkw = lambda k, ci, co: (torch.randn((k, k, ci, co), generator=g) * (2.0 / (k * k * ci)) ** 0.5).to(dev)
    h, w = 24, 78
    names = ["entry", "issued", "tile_lds", "stage0", "e1_done", "e3_done", "drained", "end"]
    print("all times in microseconds; per stamp: median / max over the launch's workgroups, relative to the FIRST workgroup's entry")
    for i, (name, cin, s, e) in enumerate(LATE):
        s2 = LATE[i + 1][2] if i + 1 < len(LATE) else 0
        sq = torch.randn((a.batch, h, w, s), generator=g).clamp_(min=0).to(dev, dt)
        w1, w3 = mkw(1, s, e), mkw(3, s, e)
        wn = mkw(1, 2 * e, s2) if s2 else None
        bz = [torch.zeros(n_, device=dev) for n_ in (e, e, max(s2, 1))]
        chain = ops.FireChainStream(w1, w3, wn, dt)
        fn = lambda: ops.fire_chain(sq, chain, bz[0], bz[1], bz[2] if s2 else None, want_y=not s2)
        ms = timeit(fn, a.iters)
        torch.cuda.synchronize()
        nwg = ((a.batch + 1) // 2) * ((w + 15) // 16) * ((h + 7) // 8)
        grid = (nwg + 7) // 8 * 8
        buf = (C.c_ulonglong * (grid * 8))()
        assert lib.sqdet_debug_chain_timeline(buf, grid * 8) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(grid, 8).astype(np.float64)
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        rel = (t - t0) / 100.0
        print("%s -> %s  S=%d E=%d S2=%d: %d workgroups, launch %.2f us (events, graph of %d), first entry -> last end %.2f us"
              % (name, LATE[i + 1][0] if s2 else "concat", s, e, s2, len(t), ms * 1e3, a.iters, rel[:, 7].max()))
        print("   " + "  ".join("%s %.2f/%.2f" % (n_, np.median(rel[:, k]), rel[:, k].max()) for k, n_ in enumerate(names)))
        d = np.diff(rel, axis=1)
        print("   phases (median per workgroup): " + "  ".join("%s->%s %.2f" % (names[k], names[k + 1], np.median(d[:, k])) for k in range(7)))


if __name__ == "__main__":
    main()
