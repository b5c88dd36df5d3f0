#!/usr/bin/env python
"""A/B of the four expand(+pool)+next-squeeze launches of the SqueezeDet forward at batch 32 (375x1242 maps): the DMA-fed kernel
(fire3.hip) against fire_stream's forms ("dbg" 70) and its own alternative geometries ("dbg" 71..), interleaved on one box, outputs
compared bitwise with the "dbg" 70 result.
    python tools/exp_firedma.py [dbg values, default "70 0"] [--reps 30]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402

FORMS = [("fire2", 94, 311, 16, 64, 16, False), ("fire3+pool3", 94, 311, 16, 64, 32, True),
         ("fire4", 47, 156, 32, 128, 32, False), ("fire5+pool5", 47, 156, 32, 128, 48, True)]


def main():
    argv = [a for a in sys.argv[1:]]
    reps = 30
    if "--reps" in argv:
        i = argv.index("--reps")
        reps = int(argv[i + 1])
        del argv[i:i + 2]
    dbgs = [int(a) for a in argv] or [70, 0]
    dev = "cuda:0"
    rs = np.random.RandomState(0)
    mk = lambda k, ci, co: torch.from_numpy((rs.randn(k, k, ci, co) * (2.0 / (k * k * ci)) ** 0.5).astype(np.float32)).to(dev)
    for name, h, w, s, e, s2, pool in FORMS:
        p1, p3, pn = [ops.pack_conv_weights(x, torch.float16) for x in (mk(1, s, e), mk(3, s, e), mk(1, 2 * e, s2))]
        b1, b3, bn = [torch.from_numpy(rs.uniform(-0.3, 0.3, c).astype(np.float32)).to(dev) for c in (e, e, s2)]
        sqs = [torch.relu(torch.randn(32, h, w, s, device=dev)).half() for _ in range(4)]
        fn = lambda k: ops.fire_expand_squeeze_next(sqs[k % 4], p1, b1, p3, b3, pn, bn, pool=pool)
        ops.set_option("dbg", 70)
        want = fn(0).clone()
        times = {d: [] for d in dbgs}
        same = {}
        for d in dbgs:
            ops.set_option("dbg", d)
            got = fn(0)
            torch.cuda.synchronize()
            same[d] = bool(torch.equal(got, want))
        for r in range(reps):
            for d in dbgs:
                ops.set_option("dbg", d)
                fn(r)
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                for k in range(4):
                    fn(r + k + 1)
                en.record()
                torch.cuda.synchronize()
                times[d].append(st.elapsed_time(en) * 1e3 / 4)
        ops.set_option("dbg", 0)
        print("%-12s " % name + "  ".join("dbg %3d: %6.1f us (min %6.1f) %s" % (d, float(np.median(times[d])), float(np.min(times[d])),
                                                                                 "bitwise" if same[d] else "DIFFERS") for d in dbgs), flush=True)


if __name__ == "__main__":
    main()
