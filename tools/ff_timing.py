#!/usr/bin/env python
"""Where a fire_fused workgroup's time goes (experiment): needs libsqdet_hip.so with fire.hip compiled
-DSQDET_FIRE_TIMING.  One launch of a late fire module at batch 32; mean s_memtime ticks per wave and segment.
    python tools/ff_timing.py fire10"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import _lib, ops  # noqa: E402

SHAPES = {"fire6": (256, 48, 192), "fire7": (384, 48, 192), "fire8": (384, 64, 256), "fire9": (512, 64, 256), "fire10": (512, 96, 384), "fire11": (768, 96, 384)}
SEG = ["A: prologue (first loads)", "A: wait + barrier", "A: MFMAs + next issue", "A: bias/relu/LDS store", "barrier A->B", "B: K loops (MFMA)", "B: fill + epilogue", "-"]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "fire10"
    cin, s, e = SHAPES[name]
    h, w = 24, 78
    dev = "cuda:0"
    rs = np.random.RandomState(0)
    mk = lambda k, ci, co: torch.from_numpy((rs.randn(k, k, ci, co) * (2.0 / (k * k * ci)) ** 0.5).astype(np.float32)).to(dev)
    ps, p1, p3 = [ops.pack_conv_weights(x, torch.float16) for x in (mk(1, cin, s), mk(1, s, e), mk(3, s, e))]
    bs, b1, b3 = [torch.zeros(c, device=dev) for c in (s, e, e)]
    x = torch.randn(32, h, w, cin, device=dev).half()
    fn = lambda: ops.fire(x, ps, bs, p1, b1, p3, b3)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record(); fn(); en.record(); torch.cuda.synchronize()
    lib = _lib.lib()
    n = 2048 * 8
    buf = (C.c_ulonglong * n)()
    lib.sqdet_debug_ff_timing.argtypes = [C.c_void_p, C.c_int]
    assert lib.sqdet_debug_ff_timing(buf, n) == 0
    t = np.array(buf[:], dtype=np.float64).reshape(2048, 8)
    t = t[t.sum(1) > 0]
    print("%s: %.1f us, %d waves; s_memtime ticks per wave:" % (name, st.elapsed_time(en) * 1e3, len(t)))
    for k in range(7):
        print("  %-28s %8.0f   (min %6.0f max %6.0f)" % (SEG[k], t[:, k].mean(), t[:, k].min(), t[:, k].max()))
    print("  %-28s %8.0f" % ("total", t.sum(1).mean()))


if __name__ == "__main__":
    main()


def placement():
    """Which workgroups share a CU, and when they start (slot 7 of the timing record)."""
    lib = _lib.lib()
    n = 2048 * 8
    buf = (C.c_ulonglong * n)()
    lib.sqdet_debug_ff_timing.argtypes = [C.c_void_p, C.c_int]
    assert lib.sqdet_debug_ff_timing(buf, n) == 0
    t = np.array(buf[:], dtype=np.uint64).reshape(2048, 8)
    import collections
    cus = collections.defaultdict(list)
    for w in range(0, 2048, 4):          # wave 0 of workgroup w / 4
        v = int(t[w, 7])
        if v == 0:
            continue
        start, hw, xcc = v & 0xFFFFFFFFFF, (v >> 40) & 0xFFFF, (v >> 56) & 0xF
        cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
        cus[(xcc, se, sh, cu)].append((w // 4, start))
    cnt = collections.Counter(len(v) for v in cus.values())
    print("CUs used: %d; workgroups per CU histogram: %s" % (len(cus), dict(cnt)))
    t0 = min(s for v in cus.values() for _, s in v)
    pairs = [sorted(v, key=lambda e: e[1]) for v in cus.values() if len(v) >= 2]
    print("first 12 CUs with >= 2 workgroups: (block id, start tick) ...")
    for v in pairs[:12]:
        print("   ", [(b, s - t0) for b, s in v])
    d = [v[1][0] - v[0][0] for v in pairs]
    print("block-id distance of co-resident pairs: min %d max %d; equal to 256: %d of %d" % (min(d), max(d), sum(1 for x in d if abs(x) == 256), len(d)))


if __name__ == "__main__" and "--placement" in sys.argv:
    placement()
