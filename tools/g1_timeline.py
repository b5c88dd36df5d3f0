#!/usr/bin/env python
"""Where a conv1x1_pipe launch's time goes, per workgroup (experiment): needs a library built with gemm1x1.hip compiled
-DSQDET_G1_TIMELINE:
    SQDET_BUILD_SUFFIX=_tl SQDET_EXTRA_DEFINES="-DSQDET_G1_TIMELINE" python -m squeezedet_amd.build
    gpurun -- 'SQDET_LIB=$GRAFT_REPO_ROOT/squeezedet_amd/libsqdet_hip_tl.so python tools/g1_timeline.py'
Seven 100 MHz s_memrealtime stamps per workgroup of the LAST of ITERS launches (rotating inputs), relative to the first workgroup's entry:
   entry | first NS-1 chunks requested | chunk 0 landed | K loop done | queue drained | last store issued | stores retired"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import _lib, ops  # noqa: E402
from tools.ab_conv1x1_shapes import SHAPES  # noqa: E402

DEV = "cuda:0"


def main():
    lib = _lib.lib()
    if not hasattr(lib, "sqdet_debug_g1_timeline"):
        sys.exit("gemm1x1.hip was not compiled with -DSQDET_G1_TIMELINE")
    lib.sqdet_debug_g1_timeline.argtypes = [C.c_void_p, C.c_int]
    names = ["entry", "requested", "chunk0", "loop", "drained", "stored", "retired"]
    dbg = int(os.environ.get("G1_DBG", "51"))     # 61 / 62 / 63: the experiment forms of the K loop (gemm1x1.hip, P1Args::dbg)
    ops.set_option("dbg", dbg)
    print("all times in microseconds; per stamp: median / max over the launch's workgroups, relative to the FIRST workgroup's entry")
    only = sys.argv[1:]
    probes = [("probe 1 workgroup", 1, 1, 64, 1024, 256, 0), ("probe 32 workgroups", 1, 32, 64, 1024, 256, 0)]
    for name, n, h, w, cin, cout, add in SHAPES + probes:
        if only and not any(o in name for o in only):
            continue
        rs = np.random.RandomState(0)
        in_bytes = n * h * w * cin * 2
        nrot = max(2, int(np.ceil(1.3 * (256 << 20) / in_bytes)))
        base = torch.from_numpy(np.maximum(rs.randn(n, h, w, cin), 0).astype(np.float16)).to(DEV)
        xs = [base.clone() for _ in range(nrot)]
        pk = ops.pack_conv_weights(torch.from_numpy((rs.randn(1, 1, cin, cout) * 0.05).astype(np.float32)).to(DEV), torch.float16)
        b = torch.zeros(cout, dtype=torch.float32, device=DEV)
        y = torch.zeros((n, h, w, cout), dtype=torch.float16, device=DEV)
        spin = int(os.environ.get("G1_SPIN", "4000"))       # launches before the one that is read: the clocks ramp over ~100 ms
        for i in range(spin):
            ops.conv2d_nhwc(xs[i % nrot], pk, b, 1, "SAME", True, out=y, accumulate=bool(add))
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * (8192 * 8))()
        assert lib.sqdet_debug_g1_timeline(buf, 8192 * 8) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)
        t = t[t[:, 6] > 0]
        t = t[t[:, 0] > t[:, 0].max() - 1e5]
        if dbg == 71:
            print("%s: shader cycles per workgroup (wave 0) in the K loop's regions: first-half MFMAs + fragment loads %.0f | wait + barrier %.0f | "
                  "LDS reads + pieces + second-half MFMAs %.0f  (%d chunks)" % (name, np.median(t[:, 1]), np.median(t[:, 2]), np.median(t[:, 3]), (cin * 2 + 63) // 64))
            continue
        mhz = np.median(t[:, 7] / np.maximum(t[:, 6] - t[:, 0], 1.0)) * 100.0
        t = t[:, :7]      # the last launch only (stale rows of larger earlier grids are older)
        t0 = t[:, 0].min()
        rel = (t - t0) / 100.0
        print("%s  %d -> %d, %d px%s: %d workgroups, first entry -> last retired %.2f us" % (name, cin, cout, n * h * w, " add" if add else "", len(t), rel[:, 6].max()))
        print("   " + "  ".join("%s %.2f/%.2f" % (n_, np.median(rel[:, k]), rel[:, k].max()) for k, n_ in enumerate(names)))
        d = np.diff(rel, axis=1)
        print("   phases (median / p90 per workgroup): " + "  ".join("%s->%s %.2f/%.2f" % (names[k], names[k + 1], np.median(d[:, k]), np.percentile(d[:, k], 90)) for k in range(6)))
        q = np.percentile(rel[:, 0], [10, 50, 90, 100])
        print("   effective shader clock (s_memtime cycles / s_memrealtime, median): %.0f MHz" % mhz)
        print("   entry times p10 / p50 / p90 / max: %.2f %.2f %.2f %.2f" % tuple(q))
    ops.set_option("dbg", 0)


if __name__ == "__main__":
    main()
