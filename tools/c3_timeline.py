#!/usr/bin/env python
"""Where a conv3x3_tile launch's time goes, per workgroup (experiment): needs a library built with -DSQDET_C3_TIMELINE:
    SQDET_BUILD_SUFFIX=_tl SQDET_EXTRA_DEFINES="-DSQDET_C3_TIMELINE" python -m squeezedet_amd.build
    gpurun -- 'SQDET_LIB=$GRAFT_REPO_ROOT/squeezedet_amd/libsqdet_hip_tl.so python tools/c3_timeline.py'
100 MHz s_memrealtime stamps per workgroup of the LAST launch: entry | first stage requested | landed | K loop done | epilogue issued |
stores retired, and the time spent waiting for the later stages' tiles."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [("plus fire4 e3", 8, 92, 309, 192, 128), ("plus fire8 e3", 8, 45, 153, 384, 256), ("plus fire6 e3", 8, 45, 153, 288, 192),
          ("plus fire9 e3", 8, 22, 76, 384, 256), ("res3 2b", 8, 47, 156, 128, 128), ("res4 2b", 8, 24, 78, 256, 256)]


def main():
    lib = _lib.lib()
    if not hasattr(lib, "sqdet_debug_c3_timeline"):
        sys.exit("conv3x3.hip was not compiled with -DSQDET_C3_TIMELINE")
    lib.sqdet_debug_c3_timeline.argtypes = [C.c_void_p, C.c_int]
    N = 16384
    for name, n, h, w, cin, cout in SHAPES:
        rs = np.random.RandomState(0)
        nrot = max(2, int(np.ceil(1.3 * (256 << 20) / (n * h * w * cin * 2))))
        base = torch.from_numpy(np.maximum(rs.randn(n, h, w, cin), 0).astype(np.float16)).to(DEV)
        xs = [base.clone() for _ in range(nrot)]
        pk = ops.pack_conv_weights(torch.from_numpy((rs.randn(3, 3, cin, cout) * 0.05).astype(np.float32)).to(DEV), torch.float16)
        b = torch.zeros(cout, dtype=torch.float32, device=DEV)
        y = torch.zeros((n, h, w, cout), dtype=torch.float16, device=DEV)
        for i in range(1500):
            ops.conv2d_nhwc(xs[i % nrot], pk, b, 1, "SAME", True, out=y)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * (N * 8))()
        assert lib.sqdet_debug_c3_timeline(buf, N * 8) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(N, 8).astype(np.float64)
        t = t[t[:, 6] > 0]
        t = t[t[:, 0] > t[:, 0].max() - 2e4]          # the last launch only (200 us window)
        t0 = t[:, 0].min()
        rel = (t[:, [0, 1, 2, 3, 4, 6]] - t0) / 100.0
        names = ["entry", "requested", "landed", "loop", "epilogue", "retired"]
        print("%s %d -> %d, %d px: %d workgroups, first entry -> last retired %.2f us" % (name, cin, cout, n * h * w, len(t), rel[:, 5].max()))
        d = np.diff(rel, axis=1)
        print("   phases, median / p90 per workgroup (us): " + "  ".join("%s->%s %.2f/%.2f" % (names[k], names[k + 1], np.median(d[:, k]), np.percentile(d[:, k], 90)) for k in range(5)))
        print("   of the K loop: waiting for the later stages' tiles %.2f/%.2f us; workgroup life median %.2f us" % (
            np.median(t[:, 5]) / 100.0, np.percentile(t[:, 5], 90) / 100.0, np.median(rel[:, 5] - rel[:, 0])))
        q = np.percentile(rel[:, 0], [10, 50, 90, 100])
        print("   entry times p10 / p50 / p90 / max: %.2f %.2f %.2f %.2f" % tuple(q))


if __name__ == "__main__":
    main()
