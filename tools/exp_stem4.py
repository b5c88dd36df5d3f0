#!/usr/bin/env python
"""Experiment (round 3): the phase stem (stem4.hip) against the persistent strip-lane stem (stem3.hip, stem_algo=3):
bitwise comparison at several shapes (plain and squeeze forms) and timing at batch 32 with inputs rotating over 357 MB."""
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


def run(x, algo, sq, dbg=0):
    ops.set_option("stem_algo", algo)
    ops.set_option("dbg", dbg)
    try:
        if sq:
            return ops.stem_conv_pool_squeeze(x, pk, b, pks, bs, "SAME", "SAME")
        return ops.stem_conv_pool(x, pk, b, "SAME", "SAME")
    finally:
        ops.set_option("stem_algo", 0)
        ops.set_option("dbg", 0)


ok = True
for (n, h, wd) in [(2, 375, 1242), (1, 384, 1248), (3, 97, 600), (1, 64, 1000), (2, 31, 524), (1, 200, 2050)]:
    x = torch.from_numpy((rs.randint(0, 256, (n, h, wd, 3)) - 110.0).astype(np.float32)).to(DEV, torch.float16)
    for sq in (False, True):
        ref = run(x, 3, sq)
        for dbg in (0,):
            got = run(x, 4, sq, dbg)
            torch.cuda.synchronize()
            same = torch.equal(got, ref)
            ok &= same
            if not same:
                d = (got.float() - ref.float()).abs()
                bad = (d > 0).nonzero()
                print("MISMATCH n=%d %dx%d sq=%s dbg=%d: %d elements differ, max %g, first at %s" % (n, h, wd, sq, dbg, int((d > 0).sum()), float(d.max()), bad[0].tolist()))
print("bitwise equal to stem_pers on all shapes:", ok)

n, h, wd = 32, 375, 1242
xs = [torch.from_numpy((rs.randint(0, 256, (n, h, wd, 3)) - 110.0).astype(np.float32)).to(DEV, torch.float16) for _ in range(4)]
for name, algo, dbg in (("stem_pers (stem3)", 3, 0), ("stem_phase 2/CU", 4, 0), ("phase no stores", 4, 101), ("phase no loads", 4, 102), ("phase neither", 4, 103), ("pers no stores", 3, 101), ("pers no loads", 3, 102), ("pers neither", 3, 103)):
    for sq in (True, False):
        for i in range(3):
            run(xs[i % 4], algo, sq, dbg)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(3):
            ops.set_option("stem_algo", algo); ops.set_option("dbg", dbg)
            st.record()
            for i in range(20):
                if sq:
                    ops.stem_conv_pool_squeeze(xs[i % 4], pk, b, pks, bs, "SAME", "SAME")
                else:
                    ops.stem_conv_pool(xs[i % 4], pk, b, "SAME", "SAME")
            en.record()
            en.synchronize()
            best = min(best, st.elapsed_time(en) / 20 * 1e3)
        ops.set_option("stem_algo", 0); ops.set_option("dbg", 0)
        print("%-20s %-8s %7.2f us" % (name, "squeeze" if sq else "plain", best))
