#!/usr/bin/env python
"""Where a stem_strip workgroup's time goes (experiment): needs libsqdet_hip.so with stem2.hip compiled
-DSQDET_FIRE_TIMING.  conv1 + pool1 of SqueezeDet at batch 32, 375x1242; s_memtime ticks per wave and segment
(the first 512 workgroups)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import _lib, ops  # noqa: E402

dev = "cuda:0"
rs = np.random.RandomState(0)
w = torch.from_numpy((rs.randn(3, 3, 3, 64) * 0.2).astype(np.float32)).to(dev)
pw = ops.pack_conv_weights(w, torch.float16)
b = torch.zeros(64, device=dev)
x = torch.randn(32, 375, 1242, 3, device=dev).half()
fn = lambda: ops.stem_conv_pool(x, pw, b)
for _ in range(3):
    fn()
torch.cuda.synchronize()
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record(); fn(); en.record(); torch.cuda.synchronize()
lib = _lib.lib()
n = 2048 * 8
buf = (C.c_ulonglong * n)()
lib.sqdet_debug_stem_timing.argtypes = [C.c_void_p, C.c_int]
assert lib.sqdet_debug_stem_timing(buf, n) == 0
t = np.array(buf[:], dtype=np.float64).reshape(2048, 8)
t = t[t.sum(1) > 0]
print("stem: %.1f us, %d waves recorded; s_memtime ticks per wave:" % (st.elapsed_time(en) * 1e3, len(t)))
for k, name in enumerate(["stage input patch (loads -> LDS)", "barrier", "conv + pool + stores"]):
    print("  %-34s %8.0f   (min %6.0f max %6.0f)" % (name, t[:, k].mean(), t[:, k].min(), t[:, k].max()))
print("  %-34s %8.0f" % ("total", t.sum(1).mean()))
