#!/usr/bin/env python
"""The fire modules' 1x1 convs (squeeze1x1, expand1x1: north_star's ">= 60 % of HBM peak on the fire-module 1x1 convs") as
STAND-ALONE launches of sqdet_conv2d_nhwc_fwd at BASELINE.json's configs[1] shapes (batch 32, 375x1242, float16), every
launch reading a DIFFERENT input copy from a rotation larger than the 256 MiB Infinity Cache.  In the benchmarked plan these
convs are not launches (a module is one fused kernel); this is the unfused path (training forward).

    python tools/fire1x1_standalone.py [--plan plan.json]                       HIP-event timing, prints the table
    rocprofv3 --kernel-trace --stats -d D -o t --output-format csv -- python tools/fire1x1_standalone.py --plan D/plan.json
    rocprofv3 --pmc FETCH_SIZE -d D/f ... ; rocprofv3 --pmc WRITE_SIZE -d D/w ...   (separate passes)
    python tools/fire1x1_standalone.py --summarize D/plan.json <kernel_trace.csv> [<fetch csv> <write csv>] > profiles/rNN_fire_1x1_standalone.txt
"""
import argparse
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FIRES = [("fire2", 64, 16, 64), ("fire3", 128, 16, 64), ("fire4", 128, 32, 128), ("fire5", 256, 32, 128),
         ("fire6", 256, 48, 192), ("fire7", 384, 48, 192), ("fire8", 384, 64, 256), ("fire9", 512, 64, 256),
         ("fire10", 512, 96, 384), ("fire11", 768, 96, 384)]
WARM, ITERS = 100, 20      # (100: the clocks ramp over the first milliseconds of a burst)
HBM_PEAK = 8000.0


def layer_list(h, w):
    o = lambda n: -(-n // 2)
    h, w = o(o(h)), o(o(w))                 # after conv1 (s2) and pool1 (s2)
    out = []
    for name, cin, s, e in FIRES:
        out.append((name + "/squeeze1x1", h, w, cin, s))
        out.append((name + "/expand1x1", h, w, s, e))
        if name in ("fire3", "fire5"):
            h, w = o(h), o(w)
    return out


def run(args):
    import numpy as np
    import torch
    from squeezedet_amd import ops
    dev = "cuda:0"
    plan = []
    rs = np.random.RandomState(0)
    print("# batch %d, %dx%d, float16; %d launches per layer after %d warm-up launches, inputs rotate over > 333 MB" % (args.batch, args.width, args.height, ITERS, WARM))
    print("%-22s %9s %9s %8s %8s" % ("layer", "us/launch", "alg MB", "GB/s", "of 8TB/s"))
    for name, h, w, cin, cout in layer_list(args.height, args.width):
        in_bytes = args.batch * h * w * cin * 2
        nrot = max(2, int(np.ceil(1.3 * (256 << 20) / in_bytes)))
        xs = [torch.from_numpy(np.maximum(rs.randn(args.batch, h, w, cin), 0).astype(np.float16)).to(dev) for _ in range(min(nrot, 2))]
        while len(xs) < nrot:
            xs.append(xs[len(xs) % 2].clone())
        wk = torch.from_numpy((rs.randn(1, 1, cin, cout) * 0.1).astype(np.float32)).to(dev)
        b = torch.zeros(cout, dtype=torch.float32, device=dev)
        pk = ops.pack_conv_weights(wk, torch.float16)
        y = torch.empty((args.batch, h, w, cout), dtype=torch.float16, device=dev)
        for i in range(WARM):
            ops.conv2d_nhwc(xs[i % nrot], pk, b, 1, "SAME", True, out=y)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for i in range(ITERS):
            ops.conv2d_nhwc(xs[(WARM + i) % nrot], pk, b, 1, "SAME", True, out=y)
        en.record()
        en.synchronize()
        us = st.elapsed_time(en) / ITERS * 1e3
        alg = (args.batch * h * w * (cin + cout) + cin * cout) * 2 + 4 * cout
        # the size bound: a plain copy kernel (sqdet_copy_channels) moving the SAME number of bytes -- a [pixels, 64] float16
        # tensor of (in + out) / 2 bytes read and written once, sources rotating the same way
        cp_pix = (in_bytes + args.batch * h * w * cout * 2) // 2 // 128
        csrc = [torch.empty((cp_pix, 64), dtype=torch.float16, device=dev).zero_() for _ in range(max(2, int(np.ceil(1.3 * (256 << 20) / (cp_pix * 128)))))]
        cdst = torch.empty((cp_pix, 64), dtype=torch.float16, device=dev)
        for i in range(WARM):
            ops.copy_channels(csrc[i % len(csrc)], cdst, 0)
        torch.cuda.synchronize()
        st.record()
        for i in range(ITERS):
            ops.copy_channels(csrc[(WARM + i) % len(csrc)], cdst, 0)
        en.record()
        en.synchronize()
        cp_us = st.elapsed_time(en) / ITERS * 1e3
        del csrc, cdst
        plan.append(dict(layer=name, launches=WARM + ITERS, warm=WARM, alg_bytes=alg, h=h, w=w, cin=cin, cout=cout, event_us=us,
                         copy_us=cp_us, copy_bytes=cp_pix * 256))
        print("%-22s %9.2f %9.1f %8.0f %8.3f   copy of the same bytes: %6.2f us = %5.0f GB/s" % (name, us, alg / 1e6, alg / us / 1e3, alg / us / 1e3 / HBM_PEAK,
                                                                                             cp_us, cp_pix * 256 / cp_us / 1e3))
        del xs, y
        torch.cuda.empty_cache()
    if args.plan:
        json.dump(plan, open(args.plan, "w"), indent=1)


def summarize(plan_path, trace_csv, fetch_csv=None, write_csv=None):
    plan = json.load(open(plan_path))
    def conv_rows(path, key):
        rows = list(csv.DictReader(open(path)))
        rows.sort(key=lambda r: int(r[key]))
        return [r for r in rows if "conv1x1" in r["Kernel_Name"]]
    tr = conv_rows(trace_csv, "Start_Timestamp")
    need = sum(p["launches"] for p in plan)
    assert len(tr) == need, "kernel trace has %d conv1x1 dispatches, the plan %d" % (len(tr), need)
    pmc = None
    if fetch_csv and write_csv:
        def load(path, counter):
            rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "conv1x1" in r["Kernel_Name"]]
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            return [float(r["Counter_Value"]) for r in rows]
        f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
        assert len(f) == need and len(w) == need, (len(f), len(w), need)
        pmc = (f, w)
    print("# The fire modules' 1x1 convs as STAND-ALONE launches (sqdet_conv2d_nhwc_fwd: conv1x1_stream for K <= 4 chunks, the LDS-resident-weights")
    print("# streaming kernel conv1x1_deepk beyond), batch 32, 375x1242, float16, one MI355X.  us = rocprofv3 --kernel-trace mean over the %d launches behind %d warm-up" % (ITERS, WARM))
    print("# ones, every launch on a different input copy (rotation > 333 MB: nothing survives in the 256 MiB Infinity Cache);")
    print("# alg = input + output + weights, each touched once; traffic = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes) from separate")
    print("# rocprofv3 --pmc passes of the same command (gfx950 correction, MI355X_MICROARCH.md); frac = alg GB/s / 8000.")
    print("# In the benchmarked plan these convs are not launches: a module is one fused kernel (DESIGN.md section 3).")
    print("# copy = sqdet_copy_channels moving the same number of bytes (HIP events, same rotation): what a plain streaming kernel")
    print("# reaches at this launch SIZE -- below ~100 MB a launch is ramp + tail, not bandwidth.")
    print("%-22s %-34s %8s %8s %8s %7s %10s %8s %9s %9s" % ("layer", "kernel", "us", "alg MB", "GB/s", "frac", "traffic MB", "traf/alg", "copy us", "copy GB/s"))
    i = 0
    for p in plan:
        rows = tr[i + p["warm"]: i + p["launches"]]
        us = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows) / len(rows) / 1e3
        kn = rows[0]["Kernel_Name"]
        kn = kn[kn.find("conv1x1"):][:34]
        traf = ""
        ratio = ""
        if pmc:
            f = sum(pmc[0][i + p["warm"]: i + p["launches"]]) / len(rows)
            w = sum(pmc[1][i + p["warm"]: i + p["launches"]]) / len(rows)
            tb = 2 * f * 1024 + w * 1024
            traf, ratio = "%10.1f" % (tb / 1e6), "%8.2f" % (tb / p["alg_bytes"])
        print("%-22s %-34s %8.2f %8.1f %8.0f %7.3f %10s %8s %9.2f %9.0f" % (p["layer"], kn, us, p["alg_bytes"] / 1e6, p["alg_bytes"] / us / 1e3,
                                                                            p["alg_bytes"] / us / 1e3 / HBM_PEAK, traf, ratio, p.get("copy_us", 0.0),
                                                                            p.get("copy_bytes", 0) / max(p.get("copy_us", 1.0), 1e-9) / 1e3))
        i += p["launches"]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--plan", default="")
    ap.add_argument("--summarize", nargs="+", default=None)
    ap.add_argument("--opt", action="append", default=[], help="tuning knob name=value (sqdet_set_option), repeatable -- A/B")
    a = ap.parse_args()
    if a.opt:
        from squeezedet_amd import ops as _ops
        for o in a.opt:
            k, v = o.split("=")
            _ops.set_option(k, int(v))
    if a.summarize:
        summarize(*a.summarize)
    else:
        run(a)
