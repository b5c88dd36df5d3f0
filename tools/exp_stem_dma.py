#!/usr/bin/env python
"""Experiment (round 4): the DMA-fed forms of the phase stem (stem4.hip stem_phase_dma, "stem_algo" 5..8) against the register-prefetch
form ("stem_algo" 4) and stem_pers (3): bitwise comparison at several shapes (plain and squeeze forms), timing at batch 32 with inputs
rotating over 357 MB, interleaved."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402

DEV = "cuda:0"
rs = np.random.RandomState(0)
w = torch.from_numpy((rs.randn(3, 3, 3, 64) * (2.0 / 27) ** 0.5 / 64).astype(np.float32)).to(DEV)
b = torch.from_numpy(rs.uniform(-0.5, 0.5, 64).astype(np.float32)).to(DEV)
ws = torch.from_numpy((rs.randn(1, 1, 64, 16) * 0.2).astype(np.float32)).to(DEV)
bs = torch.from_numpy(rs.uniform(-0.1, 0.1, 16).astype(np.float32)).to(DEV)
pk, pks = ops.pack_conv_weights(w, torch.float16), ops.pack_conv_weights(ws, torch.float16)
ALGOS = [int(a) for a in sys.argv[1:]] or [4, 5, 6, 7, 8]


def run(x, algo, sq):
    ops.set_option("stem_algo", algo)
    try:
        if sq:
            return ops.stem_conv_pool_squeeze(x, pk, b, pks, bs, "SAME", "SAME")
        return ops.stem_conv_pool(x, pk, b, "SAME", "SAME")
    finally:
        ops.set_option("stem_algo", 0)


ok = True
for (n, h, wd) in [(2, 375, 1242), (1, 384, 1248), (3, 97, 600), (1, 64, 1000), (2, 31, 524), (1, 200, 2050), (2, 33, 1030)]:
    x = torch.from_numpy((rs.randint(0, 256, (n, h, wd, 3)) - 110.0).astype(np.float32)).to(DEV, torch.float16)
    for sq in (False, True):
        ref = run(x, 3, sq)
        for algo in ALGOS:
            got = run(x, algo, sq)
            torch.cuda.synchronize()
            same = torch.equal(got, ref)
            ok &= same
            if not same:
                d = (got.float() - ref.float()).abs()
                bad = (d > 0).nonzero()
                print("MISMATCH algo %d n=%d %dx%d sq=%s: %d elements differ, max %g, first at %s last at %s" % (algo, n, h, wd, sq, int((d > 0).sum()), float(d.max()), bad[0].tolist(), bad[-1].tolist()))
print("bitwise equal to stem_pers on all shapes:", ok)

n, h, wd = 32, 375, 1242
xs = [torch.from_numpy((rs.randint(0, 256, (n, h, wd, 3)) - 110.0).astype(np.float32)).to(DEV, torch.float16) for _ in range(4)]
times = {a: [] for a in ALGOS}
for rep in range(6):
    for algo in ALGOS:
        ops.set_option("stem_algo", algo)
        ops.stem_conv_pool_squeeze(xs[0], pk, b, pks, bs, "SAME", "SAME")
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for i in range(20):
            ops.stem_conv_pool_squeeze(xs[i % 4], pk, b, pks, bs, "SAME", "SAME")
        en.record()
        en.synchronize()
        times[algo].append(st.elapsed_time(en) / 20 * 1e3)
ops.set_option("stem_algo", 0)
for a in ALGOS:
    print("stem_algo %d (squeeze form): median %.2f us  min %.2f" % (a, float(np.median(times[a])), min(times[a])))
