#!/usr/bin/env python
"""A/B of the ConvDet launch (768 -> 72, 3x3) at batch 32 on the 24x78 map: the DMA-staged kernel (default) against the register-prefetch form
("dbg" 80), with and without the score epilogue, bitwise comparison at several shapes, interleaved timing.
    python tools/exp_convdet.py [dbg values, default "0 80"]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402

DEV = "cuda:0"
dbgs = [int(a) for a in sys.argv[1:]] or [80, 0]
rs = np.random.RandomState(0)
w = torch.from_numpy((rs.randn(3, 3, 768, 72) * (2.0 / (9 * 768)) ** 0.5 * 2).astype(np.float32)).to(DEV)
b = torch.from_numpy(rs.uniform(-0.1, 0.1, 72).astype(np.float32)).to(DEV)
pk = ops.pack_conv_weights(w, torch.float16)
ok = True
for (n, h, wd) in [(2, 24, 78), (3, 22, 76), (1, 9, 17), (5, 8, 16), (2, 25, 33)]:
    x = torch.from_numpy(np.maximum(rs.randn(n, h, wd, 768), 0).astype(np.float32)).to(DEV, torch.float16)
    ops.set_option("dbg", 80)
    p0, s0 = ops.convdet(x, pk, b, 9, 3)
    y0 = ops.conv2d_nhwc(x, pk, b, 1, "SAME", False)
    for d in dbgs:
        ops.set_option("dbg", d)
        p, s = ops.convdet(x, pk, b, 9, 3)
        y = ops.conv2d_nhwc(x, pk, b, 1, "SAME", False)
        torch.cuda.synchronize()
        same = torch.equal(p, p0) and torch.equal(s, s0) and torch.equal(y, y0) and torch.equal(y, p)
        ok &= same
        if not same:
            print("MISMATCH dbg %d n=%d %dx%d: preds %s scores %s plain %s" % (d, n, h, wd, torch.equal(p, p0), torch.equal(s, s0), torch.equal(y, y0)))
ops.set_option("dbg", 0)
print("bitwise equal on all shapes:", ok)
xs = [torch.from_numpy(np.maximum(rs.randn(32, 24, 78, 768), 0).astype(np.float32)).to(DEV, torch.float16) for _ in range(3)]
pr = torch.empty((32, 24, 78, 72), dtype=torch.float16, device=DEV)
sc = torch.empty((32, 24 * 78 * 9), dtype=torch.float32, device=DEV)
times = {d: [] for d in dbgs}
for rep in range(8):
    for d in dbgs:
        ops.set_option("dbg", d)
        ops.convdet(xs[0], pk, b, 9, 3, preds=pr, scores=sc)
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for i in range(10):
            ops.convdet(xs[i % 3], pk, b, 9, 3, preds=pr, scores=sc)
        en.record()
        en.synchronize()
        times[d].append(st.elapsed_time(en) / 10 * 1e3)
ops.set_option("dbg", 0)
for d in dbgs:
    print("dbg %3d (score form, batch 32, 24x78): median %.2f us  min %.2f" % (d, float(np.median(times[d])), min(times[d])))
