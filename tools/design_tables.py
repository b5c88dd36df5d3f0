#!/usr/bin/env python
"""Prints the markdown tables DESIGN.md sections 3 / 5 quote, from ONE profile collection (profiles/<tag>_*), lines <= 120 columns:
    python tools/design_tables.py r05"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = lambda n: os.path.join(ROOT, "profiles", "%s_%s" % (tag, n))
HBM, MFMA = 8000.0, 2500.0
SHORT = {"conv1+pool1+fire2/squeeze1x1": "conv1+pool1+f2/sq", "fire2/expand+fire3/squeeze1x1": "f2/exp+f3/sq",
         "fire3/expand+pool3+fire4/squeeze1x1": "f3/exp+pool3+f4/sq", "fire4/expand+fire5/squeeze1x1": "f4/exp+f5/sq",
         "fire5/expand+pool5+fire6/squeeze1x1": "f5/exp+pool5+f6/sq", "fire6/expand+fire7/squeeze1x1": "f6/exp+f7/sq",
         "fire7/expand+fire8/squeeze1x1": "f7/exp+f8/sq", "fire8/expand+fire9/squeeze1x1": "f8/exp+f9/sq",
         "fire9/expand+fire10/squeeze1x1": "f9/exp+f10/sq", "fire10/expand+fire11/squeeze1x1": "f10/exp+f11/sq",
         "fire11/expand": "f11/exp", "conv12": "conv12"}


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def load_stats(path):
    out = {}
    for line in open(path):
        if line.startswith(("#", "kernel ")) or len(line) < 100:
            continue
        out[line[:88].rstrip()] = float(line[90:].split()[2])
    return out


lt = json.load(open(P("layer_table.json")))
pm = json.load(open(P("hbm_traffic_pmc.json")))
kname = {k["layer"]: k["kernel"] for k in pm["kernels"]}
stats2 = load_stats(P("kernel_stats.txt"))                       # the benchmarked step: two forwards in flight
stats = load_stats(P("kernel_stats_1lane.txt")) if os.path.exists(P("kernel_stats_1lane.txt")) else stats2
b = last_json(P("bench_sqdet_infer.json"))
b1 = last_json(P("bench_sqdet_infer_1lane.json")) if os.path.exists(P("bench_sqdet_infer_1lane.json")) else None
box = (b1 or b).get("box") or {}
box_tf = box.get("box_mfma_tflops") or MFMA
print("collection `profiles/%s_*`: build `%s`; one-lane run on a %s MHz box, `box_mfma_tflops` %s (16x16x32, real operands);" % (
    tag, lt["build_fingerprint"], (b1 or b)["clocks"]["before"].get("gfxclk_mhz"), box.get("box_mfma_tflops")))
print("us: rocprofv3 kernel trace of `bench.py`; spec = of 8 TB/s or 2.5 PF/s, box = of the box's own MFMA calibration\n")
print("| launch | us, 1 lane | us, 2 lanes | alg MB | traffic MB | GFLOP | bound | frac spec | frac box |")
print("|---|---|---|---|---|---|---|---|---|")
tot = 0.0
for l in lt["layers"]:
    k = kname[l["layer"]][:88].rstrip()
    us = stats.get(k)
    shared = sum(1 for x in kname.values() if x == kname[l["layer"]])
    inten = l["flops"] / l["bytes"]
    bound = "MFMA" if inten > MFMA * 1e12 / (HBM * 1e9) else "HBM"
    t = us if us else l["ms"] * 1e3
    if bound == "MFMA":
        fs = l["flops"] / (t * 1e-6) / 1e12 / MFMA
        fb = "%.3f" % (l["flops"] / (t * 1e-6) / 1e12 / box_tf)
    else:
        fs = l["bytes"] / (t * 1e-6) / 1e9 / HBM
        fb = "-"
    tot += t
    print("| %s | %s%s | %s | %.0f | %.0f | %.1f | %s | %.3f | %s |" % (
        SHORT.get(l["layer"], l["layer"]), "%.1f" % us if us else "-", " (mean of %d)" % shared if shared > 1 else "",
        "%.1f" % stats2[k] if k in stats2 else "-", l["bytes"] / 1e6, pm["by_layer"][l["layer"]] / 1e6, l["flops"] / 1e9, bound, fs, fb))
print("| **sum** | **%.0f** | | %.0f | %.0f | %.0f | | | |" % (tot, sum(l["bytes"] for l in lt["layers"]) / 1e6, sum(pm["by_layer"].values()) / 1e6,
                                                            sum(l["flops"] for l in lt["layers"]) / 1e9))
if b1:
    print("\none lane: step %.4f ms = %.1f k img/s; two lanes: step %.4f ms = %.1f k img/s (box %s MHz, `box_mfma_tflops` %s)" % (
        b1["ms_per_step"], b1["value"] / 1e3, b["ms_per_step"], b["value"] / 1e3, b["clocks"]["before"].get("gfxclk_mhz"),
        (b.get("box") or {}).get("box_mfma_tflops")))
if b.get("pipeline"):
    p = b["pipeline"]
    print("chip level (two lanes): %.1f TF/s = %.3f of 2.5 PF/s = %.3f of the box's MFMA calibration; composite roofline %.4f ms = %.3f of the step" % (
        p["achieved_tflops"], p["frac_of_mfma_peak"], p["achieved_tflops"] / ((b.get("box") or {}).get("box_mfma_tflops") or MFMA),
        p["composite_roofline_ms"], p["composite_roofline_frac"]))
if b.get("latency_ms_per_batch"):
    print("latency per 32-image batch: %s" % json.dumps({k: v for k, v in b["latency_ms_per_batch"].items() if k != "note"}))
print()
print("| config | value | ms/step | box MFMA TF/s, MHz | roofline object of the line |")
print("|---|---|---|---|---|")
for c in ("sqdet_infer", "sqdet_infer_1lane", "sqdet_infer_20steps", "sqdet_infer_384", "sqdet_sample_b1", "sqdetplus_infer", "sqdet_train_fp32",
          "sqdet_train_fp16", "res50_train_fp16"):
    try:
        d = last_json(P("bench_%s.json" % c))
    except OSError:
        continue
    r = d.get("roofline", {})
    note = "%s: %.3f" % (r.get("kernel", "")[:34], r.get("frac", 0))
    if r.get("rocprof_frac"):
        note += " (rocprof %.3f)" % r["rocprof_frac"]
    print("| `%s` | %.0f img/s | %.4f | %s, %s | %s |" % (c, d["value"], d["ms_per_step"], (d.get("box") or {}).get("box_mfma_tflops"),
                                                      d["clocks"]["before"].get("gfxclk_mhz"), note))
print()
print("cpu_baseline objects: " + "; ".join(
    "`%s` %.1f img/s on %d threads" % (c, last_json(P("bench_%s.json" % c))["cpu_baseline"]["value"], last_json(P("bench_%s.json" % c))["cpu_baseline"]["cores"])
    for c in ("sqdet_infer", "sqdetplus_infer", "sqdet_train_fp32", "res50_train_fp16")
    if os.path.exists(P("bench_%s.json" % c)) and last_json(P("bench_%s.json" % c)).get("cpu_baseline")))
