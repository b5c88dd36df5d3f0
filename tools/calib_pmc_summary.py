#!/usr/bin/env python
"""Joins a rocprofv3 --pmc counter_collection.csv with the kernel_trace.csv of the same run for the calib_mfma2 launches:
per dispatch the kernel duration, GRBM_GUI_ACTIVE (summed over the 8 XCDs), the effective clock GRBM_GUI_ACTIVE / 8 / duration, and
SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x GRBM_GUI_ACTIVE / 8) = the fraction of the matrix pipes' cycles that were busy.
    python tools/calib_pmc_summary.py <counter_collection.csv> <kernel_trace.csv> [un-profiled table to prepend]"""
import collections
import csv
import sys

cc = list(csv.DictReader(open(sys.argv[1])))
kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(sys.argv[2]))}
per = collections.OrderedDict()
for r in cc:
    if "calib_mfma" not in r["Kernel_Name"]:
        continue
    d = per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"])})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
if len(sys.argv) > 3:
    print("== un-profiled (events + in-kernel s_memtime / s_memrealtime), tools/calib_run.py ==")
    print(open(sys.argv[3]).read())
print("== rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -- python tools/calib_run.py --reps 1 ==")
print("(profiled passes clock lower than un-profiled ones: MI355X_MICROARCH.md, DVFS give-back (2); dispatch order = calib_run's loop order,"
      " two dispatches per variant: warm-up + timed)")
print("%-34s %7s %10s %14s %10s %16s %10s" % ("kernel", "wgs", "dur us", "GRBM_GUI_ACT", "clock MHz", "MFMA_BUSY_CYC", "pipe busy"))
for did, d in per.items():
    k = kt.get(did)
    if not k:
        continue
    dur_ns = float(k["End_Timestamp"]) - float(k["Start_Timestamp"])
    gui = d.get("GRBM_GUI_ACTIVE", 0.0)
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    clock = gui / 8.0 / (dur_ns * 1e-3) if dur_ns > 0 else 0.0
    simds = 256 * 4
    frac = busy / (simds * gui / 8.0) if gui > 0 else 0.0
    short = "calib_mfma2<%s>" % ("32x32x16" if "Li1E" in d["name"] or "<1>" in d["name"] else "16x16x32") if "mfma2" in d["name"] else "calib_mfma (16x16x32, 2 w/SIMD)"
    print("%-34s %7d %10.1f %14.4g %10.1f %16.4g %10.4f" % (short, d["grid"] // 256, dur_ns * 1e-3, gui, clock, busy, frac))
