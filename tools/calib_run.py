#!/usr/bin/env python
"""Box calibration matrix: the bare MFMA loop in both instruction shapes, 1 / 2 / 4 waves per SIMD, non-trivial and
all-zero operands -- TF/s (events), effective shader clock and matrix-pipe busy fraction (in-kernel s_memtime / s_memrealtime).
Under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace` (tools/calib_pmc.sh) the same
launches give the counter view.    python tools/calib_run.py [--reps N] [--json out.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from squeezedet_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    rows = []
    for shape in (0, 1):
        for wps in (1, 2, 4):
            for zero in (False, True):
                rows.append(ops.calib_mfma_variant(dev, shape, wps, zero, reps=a.reps))
    print("%-9s %5s %5s %9s %8s %10s %12s %9s" % ("shape", "w/SIMD", "zero", "TF/s", "ms", "clock MHz", "cyc/MFMA/SIMD", "pipe busy"))
    for r in rows:
        print("%-9s %5d %5s %9.1f %8.4f %10.1f %12.2f %9.4f" % (r["shape"], r["waves_per_simd"], r["zero_operands"], r["tflops"], r["ms"],
                                                             r["effective_clock_mhz"], r["cycles_per_mfma_per_simd"], r["mfma_busy_frac"]))
    old = ops.box_calibration(dev)
    print("box_calibration:", json.dumps(old))
    if a.json:
        json.dump({"variants": rows, "box_calibration": old}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
