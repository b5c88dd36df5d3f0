#!/usr/bin/env python
"""Where a ConvDet (convdet_kernel, csrc/convdet.hip) workgroup's time goes (experiment): needs libsqdet_hip.so with
convdet.hip compiled -DSQDET_FIRE_TIMING (without -amdgpu-mfma-vgpr-form, as build.py does).  conv12 of SqueezeDet
(768 -> 72, 24x78) at the batch given on the command line (default 32 and 1); s_memtime ticks per wave and segment (the
first 512 workgroups; a persistent workgroup's totals cover all its tiles).  A tick is NOT a shader cycle here: 112 k
ticks per 80 us of kernel = 1.4 per ns while GRBM_GUI_ACTIVE counts 2.2-2.3 per ns -- read the segments as fractions."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import _lib, ops  # noqa: E402

dev = "cuda:0"
rs = np.random.RandomState(0)
w = torch.from_numpy((rs.randn(3, 3, 768, 72) * 0.02).astype(np.float32)).to(dev)
pw = ops.pack_conv_weights(w, torch.float16)
b = torch.zeros(72, device=dev)
lib = _lib.lib()
lib.sqdet_debug_convdet_timing.argtypes = [C.c_void_p, C.c_int]
NAMES = ["prologue (offsets, first loads)", "barrier: previous stage read", "wait for input + LDS stores", "barrier: stage visible",
         "issue next input + first B reads", "9 taps (360 MFMAs)", "K reduction + stores"]
for batch in [int(v) for v in sys.argv[1:]] or [32, 1]:
    x = torch.randn(batch, 24, 78, 768, device=dev).half()
    fn = lambda: ops.conv2d_nhwc(x, pw, b, relu=False)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record(); fn(); en.record(); torch.cuda.synchronize()
    n = 2048 * 8
    buf = (C.c_ulonglong * n)()
    assert lib.sqdet_debug_convdet_timing(buf, n) == 0
    t = np.array(buf[:], dtype=np.float64).reshape(2048, 8)[: min(2048, batch * 15 * 4)]
    t = t[t.sum(1) > 0]
    print("batch %d: conv12 %.1f us, %d waves recorded; s_memtime ticks per wave:" % (batch, st.elapsed_time(en) * 1e3, len(t)))
    for k, name in enumerate(NAMES):
        print("  %-34s %8.0f   (min %6.0f max %6.0f)" % (name, t[:, k].mean(), t[:, k].min(), t[:, k].max()))
    print("  %-34s %8.0f" % ("total", t.sum(1).mean()))
