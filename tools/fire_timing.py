#!/usr/bin/env python
"""Where a fire_stream tile goes (experiment): needs libsqdet_hip.so built with fire2.hip compiled -DSQDET_FIRE_TIMING
(see DESIGN.md).  Runs one fire module launch (batch 32 shapes of SqueezeDet) and prints the mean s_memtime cycles per
tile of each segment of the tile loop, per wave.
    python tools/fire_timing.py fire3 [--pool] [--sqnext: the module + the next module's squeeze, sqdet_fire_squeeze_next_fwd]
    [--expsq: expand (+ pool) from the squeeze tensor + next squeeze, sqdet_fire_expand_squeeze_next_fwd]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import _lib, ops  # noqa: E402

SHAPES = {"fire2": (94, 311, 64, 16, 64), "fire3": (94, 311, 128, 16, 64), "fire4": (47, 156, 128, 32, 128), "fire5": (47, 156, 256, 32, 128)}
SEG = ["A: squeeze MFMAs", "A: bias/relu/LDS store", "prefetch issue", "barrier", "B: 3x3 MFMAs", "B: 3x3 epilogue", "B: 1x1 + epilogue", "loop overhead"]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "fire3"
    pool = "--pool" in sys.argv
    for o in sys.argv[2:]:
        if "=" in o:
            k, v = o.split("=")
            ops.set_option(k, int(v))
    h, w, cin, s, e = SHAPES[name]
    dev = "cuda:0"
    rs = np.random.RandomState(0)
    mk = lambda k, ci, co: torch.from_numpy((rs.randn(k, k, ci, co) * (2.0 / (k * k * ci)) ** 0.5).astype(np.float32)).to(dev)
    ps, p1, p3 = [ops.pack_conv_weights(x, torch.float16) for x in (mk(1, cin, s), mk(1, s, e), mk(3, s, e))]
    bs, b1, b3 = [torch.zeros(c, device=dev) for c in (s, e, e)]
    x = torch.randn(32, h, w, cin, device=dev).half()
    fn = (lambda: ops.fire_maxpool(x, ps, bs, p1, b1, p3, b3)) if pool else (lambda: ops.fire(x, ps, bs, p1, b1, p3, b3))
    if "--expsq" in sys.argv:      # the module's expand half (+ pool) from its squeeze tensor + the next module's squeeze
        s2 = {"fire2": 16, "fire3": 32, "fire4": 32, "fire5": 48}[name]
        pn, bn = ops.pack_conv_weights(mk(1, 2 * e, s2), torch.float16), torch.zeros(s2, device=dev)
        sq = torch.relu(torch.randn(32, h, w, s, device=dev)).half()
        fn = lambda: ops.fire_expand_squeeze_next(sq, p1, b1, p3, b3, pn, bn, pool=pool)
    if "--sqnext" in sys.argv:
        pn, bn = ops.pack_conv_weights(mk(1, 2 * e, s), torch.float16), torch.zeros(s, device=dev)
        fn = lambda: ops.fire_squeeze_next(x, ps, bs, p1, b1, p3, b3, pn, bn)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record(); fn(); en.record(); torch.cuda.synchronize()
    lib = _lib.lib()
    n = 2048 * 8
    buf = (C.c_ulonglong * n)()
    lib.sqdet_debug_fire_timing.argtypes = [C.c_void_p, C.c_int]
    assert lib.sqdet_debug_fire_timing(buf, n) == 0
    t = np.array(buf[:], dtype=np.float64).reshape(2048, 8)
    t = t[t.sum(1) > 0]
    tiles = 32 * ((h // 2 + 3) // 4 * ((w // 2 + 6) // 7) if pool else ((h + 7) // 8) * ((w + 15) // 16))
    per_wg = tiles / (len(t) / 4.0)
    print("%s%s: %.1f us, %d waves, %.1f tiles per workgroup; s_memtime ticks per tile per wave:" % (name, "+pool" if pool else "", st.elapsed_time(en) * 1e3, len(t), per_wg))
    tot = 0.0
    for k in range(8):
        v = t[:, k].mean() / per_wg
        tot += v
        print("  %-26s %8.0f   (min %6.0f max %6.0f over waves)" % (SEG[k], v, t[:, k].min() / per_wg, t[:, k].max() / per_wg))
    print("  %-26s %8.0f" % ("total", tot))


if __name__ == "__main__":
    main()
