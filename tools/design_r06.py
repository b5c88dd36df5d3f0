#!/usr/bin/env python
"""Regenerates DESIGN.md's measured tables from ONE profile collection (profiles/<tag>_*): replaces the text between the markers
`<!-- table:NAME -->` ... `<!-- /table:NAME -->` in DESIGN.md.
    python tools/design_r06.py r06"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda n: os.path.join(ROOT, "profiles", "%s_%s" % (tag, n))


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def forward_table():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_tables.py"), tag], capture_output=True, text=True)
    if out.returncode != 0:
        return "(tools/design_tables.py failed: %s)" % out.stderr[-300:]
    return out.stdout.split("\n| config |")[0].strip()     # (its own bench table is bench_table() below)


def bench_table():
    rows = ["| config | value | ms/step | box MFMA TF/s, MHz | roofline of the dominant launch / kernel | cpu_baseline |", "|---|---|---|---|---|---|"]
    names = ["sqdet_infer", "sqdet_infer_1lane", "sqdet_infer_20steps", "sqdet_infer_384", "sqdet_sample_b1", "sqdetplus_infer", "sqdet_train_fp32",
             "sqdet_train_fp16", "res50_train_fp16"]
    for n in names:
        f = P("bench_%s.json" % n)
        if not os.path.exists(f):
            continue
        d = last_json(f)
        r = d.get("roofline") or {}
        box = d.get("box") or {}
        clk = ((d.get("clocks") or {}).get("before") or {}).get("gfxclk_mhz")
        dk = r.get("dominant_kernel")
        if isinstance(dk, dict):
            kn = re.sub(r"^_ZN5sqdet\d*(_GLOBAL__N_1\d+)?", "", str(dk.get("kernel", "")))
            kn = re.split(r"I|\(", kn)[0].lstrip("0123456789")
            roof = "step %.3f of %s peak; dominant kernel `%s`: %.1f us avg, %.0f %% of the step, %.0f TF/s = %.3f of MFMA, %.0f GB/s = %.3f of HBM" % (
                r.get("frac") or 0, r.get("bound"), kn, dk.get("avg_launch_us") or 0, 100 * (dk.get("share_of_profiled_time") or 0),
                dk.get("mfma_tflops") or 0, dk.get("mfma_frac") or 0, dk.get("hbm_gbs") or 0, dk.get("hbm_frac") or 0)
        else:
            roof = "%s: live %.1f us frac %.3f; rocprof %s us frac %s; traffic %s MB" % (
                r.get("kernel", "?"), 1e3 * (r.get("avg_launch_ms") or 0), r.get("frac") or 0,
                ("%.1f" % (1e3 * r["rocprof_avg_launch_ms"])) if r.get("rocprof_avg_launch_ms") else "-", r.get("rocprof_frac"),
                ("%.0f" % (r["traffic"] / 1e6)) if r.get("traffic") else "-")
        cb = d.get("cpu_baseline") or {}
        cpu = "%.1f %s, %s threads" % (cb["value"], cb.get("unit", ""), cb.get("cores")) if cb.get("value") else "-"
        rows.append("| `%s` | %s %s | %.4f | %s, %s | %s | %s |" % (n, ("%.0f" % d["value"]), d.get("unit", "").replace("images/s", "img/s"), d["ms_per_step"],
                                                                 box.get("box_mfma_tflops"), clk, roof, cpu))
    return "\n".join(rows)


def f1_table():
    f = P("fire_1x1_standalone.txt")
    if not os.path.exists(f):
        return "(no %s)" % f
    tabs, cur = {}, None
    for line in open(f):
        m = re.match(r"# batch (\d+)", line)
        if m:
            cur = int(m.group(1)); tabs[cur] = {}
            continue
        p = line.split()
        if cur is not None and len(p) >= 5 and "/" in p[0]:
            cp = re.search(r"bytes:\s+([\d.]+) us", line)
            tabs[cur][p[0]] = (float(p[1]), float(p[2]), float(p[4]), float(cp.group(1)) if cp else None)
    bs = sorted(tabs)
    rows = ["| layer | " + " | ".join("batch %d: alg MB, us (frac of 8 TB/s), copy us" % b for b in bs) + " |", "|---|" + "---|" * len(bs)]
    for layer in tabs[bs[0]]:
        cells = []
        for b in bs:
            us, mb, fr, cp = tabs[b].get(layer, (0, 0, 0, None))
            cells.append("%.0f, %.1f (%s%.3f%s), %s" % (mb, us, "**" if fr >= 0.6 else "", fr, "**" if fr >= 0.6 else "", ("%.1f" % cp) if cp else "-"))
        rows.append("| %s | %s |" % (layer, " | ".join(cells)))
    for b in bs:
        n = sum(1 for v in tabs[b].values() if v[2] >= 0.6)
        rows.append("")
        rows.append("batch %d: %d of %d layers at >= 0.60 of 8 TB/s." % (b, n, len(tabs[b])))
    return "\n".join(rows)


def plus_table():
    f = P("sqdetplus_infer_layer_table.json")
    if not os.path.exists(f):
        return "(no %s)" % f
    lt = json.load(open(f))
    rows = ["| launch | us (one forward in flight, events) | GFLOP | alg MB | TF/s | GB/s |", "|---|---|---|---|---|---|"]
    for l in lt["layers"]:
        rows.append("| %s | %.1f | %.1f | %.1f | %.0f | %.0f |" % (l["layer"], 1e3 * l["ms"], l["flops"] / 1e9, l["bytes"] / 1e6, l["TFLOP/s"], l["GB/s"]))
    rows.append("| **sum** | **%.0f** | | | | |" % (1e3 * lt["forward_ms_sum"]))
    return "\n".join(rows)


TABLES = {"forward": forward_table, "bench": bench_table, "fire1x1": f1_table, "plus": plus_table}


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    for name, fn in TABLES.items():
        pat = re.compile(r"(<!-- table:%s -->\n)(.*?)(<!-- /table:%s -->)" % (name, name), re.S)
        if pat.search(s):
            s = pat.sub(lambda m: m.group(1) + fn() + "\n" + m.group(3), s)
    open(path, "w").write(s)


if __name__ == "__main__":
    main()
