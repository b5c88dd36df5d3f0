#!/usr/bin/env python
"""Prints the markdown tables DESIGN.md section 3 / 5 quote, from one profile collection (profiles/<tag>_*):
    python tools/design_tables.py r04"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = lambda n: os.path.join(ROOT, "profiles", "%s_%s" % (tag, n))
HBM, MFMA = 8000.0, 2500.0
lt = json.load(open(P("layer_table.json")))
pm = json.load(open(P("hbm_traffic_pmc.json")))
kname = {k["layer"]: k["kernel"] for k in pm["kernels"]}
def load_stats(path):
    out = {}
    for line in open(path):
        if line.startswith(("#", "kernel ")) or len(line) < 100:
            continue
        out[line[:88].rstrip()] = float(line[90:].split()[2])
    return out


stats2 = load_stats(P("kernel_stats.txt"))                       # the benchmarked step: two forwards in flight
stats = load_stats(P("kernel_stats_1lane.txt")) if os.path.exists(P("kernel_stats_1lane.txt")) else stats2
b = json.loads(open(P("bench_sqdet_infer.json")).read().strip().splitlines()[-1])
b1 = json.loads(open(P("bench_sqdet_infer_1lane.json")).read().strip().splitlines()[-1]) if os.path.exists(P("bench_sqdet_infer_1lane.json")) else None
print("(`profiles/%s_*`, one collection of build `%s` on a %s MHz box, `box_mfma_tflops` %s; in-step = rocprofv3 kernel trace of"
      % (tag, lt["build_fingerprint"], b["clocks"]["before"].get("gfxclk_mhz"), (b.get("box") or {}).get("box_mfma_tflops")))
print("`bench.py`, alone = HIP events around single launches; frac = of 8 TB/s or of the 2.5 PF/s dense fp16 peak, whichever bounds the launch by intensity)\n")
print("| launch | µs in-step, one forward in flight | µs in the two-lane step (wall) | µs alone | alg MB | traffic MB | GFLOP | bound | frac (one in flight) |")
print("|---|---|---|---|---|---|---|---|---|")
tot = 0.0
for l in lt["layers"]:
    k = kname[l["layer"]][:88].rstrip()
    us = stats.get(k)
    shared = sum(1 for x in kname.values() if x == kname[l["layer"]])
    inten = l["flops"] / l["bytes"]
    bound = "MFMA" if inten > MFMA * 1e12 / (HBM * 1e9) else "HBM"
    t = us if us else l["ms"] * 1e3
    frac = (l["flops"] / (t * 1e-6) / 1e12 / MFMA) if bound == "MFMA" else (l["bytes"] / (t * 1e-6) / 1e9 / HBM)
    tot += t
    print("| %s | %s%s | %s | %.1f | %.0f | %.0f | %.1f | %s | %.3f |" % (l["layer"], "%.1f" % us if us else "—", " (mean of %d launches)" % shared if shared > 1 else "",
                                                                       "%.1f" % stats2[k] if k in stats2 else "—", l["ms"] * 1e3, l["bytes"] / 1e6,
                                                                       pm["by_layer"][l["layer"]] / 1e6, l["flops"] / 1e9, bound, frac))
print("| **sum** | **%.0f**%s | step %.4f ms = %.1f k img/s | %.0f | %.0f | %.0f | %.0f | | |" % (
    tot, " (step %.4f ms = %.1f k img/s)" % (b1["ms_per_step"], b1["value"] / 1e3) if b1 else "", b["ms_per_step"], b["value"] / 1e3,
    lt["forward_ms_sum"] * 1e3, sum(l["bytes"] for l in lt["layers"]) / 1e6, sum(pm["by_layer"].values()) / 1e6, sum(l["flops"] for l in lt["layers"]) / 1e9))
if b.get("pipeline"):
    print("\nchip-level (`pipeline` of the bench line): %s" % json.dumps(b["pipeline"]))
print()
print("| config | value | ms/step | box (MFMA TF/s, clock) | note |")
print("|---|---|---|---|---|")
for c in ("sqdet_infer", "sqdet_infer_20steps", "sqdet_infer_384", "sqdet_sample_b1", "sqdetplus_infer", "sqdet_train_fp32", "sqdet_train_fp16", "res50_train_fp16"):
    try:
        d = json.loads(open(P("bench_%s.json" % c)).read().strip().splitlines()[-1])
    except OSError:
        continue
    r = d.get("roofline", {})
    note = "%s: frac %.3f" % (r.get("kernel", "")[:40], r.get("frac", 0))
    if r.get("rocprof_frac"):
        note += " (rocprof %.3f)" % r["rocprof_frac"]
    cb = d.get("cpu_baseline")
    if cb:
        note += "; cpu_baseline %.1f img/s on %d threads (spread %s)" % (cb["value"], cb["cores"], cb.get("spread"))
    print("| `%s` | %.1f img/s | %.4f | %s, %s MHz | %s |" % (c, d["value"], d["ms_per_step"], (d.get("box") or {}).get("box_mfma_tflops"),
                                                           d["clocks"]["before"].get("gfxclk_mhz"), note))
