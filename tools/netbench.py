#!/usr/bin/env python
"""Whole-plan forward benchmark for any native arch (squeezeDet, squeezeDet+, resnet50): per-launch
milliseconds from HIP events (sqdet_net_forward_timed) with algorithmic GB/s and TFLOP/s, then the
untimed-launch throughput of sqdet_net_forward.

    python tools/netbench.py --arch resnet50 --batch 8 [--dtype fp16] [--steps 20] [--opt name=value]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from squeezedet_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet50", choices=["squeezeDet", "squeezeDet+", "resnet50"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    for kv in a.opt:
        k, v = kv.split("=")
        ops.set_option(k, int(v))
    dt = torch.float16 if a.dtype == "fp16" else torch.float32
    dev = torch.device("cuda:0")
    plan = ops.NetPlan(a.arch, dt, a.batch, a.height, a.width, 3, 9, dev)
    g = torch.Generator().manual_seed(0)
    for name, shape in plan.param_specs():
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernels":
            fan_in = shape[0] * shape[1] * shape[2]
            v = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5 / (64.0 if name == "conv1/kernels" else 1.0)
        elif leaf in ("gamma", "var"):
            v = torch.rand(shape, generator=g) * (0.3 if name.endswith("_branch2c/gamma") else 1.0) + 0.5
        else:
            v = (torch.rand(shape, generator=g) - 0.5) * 0.2
        plan.set_param(name, v.to(dev))
    x = (torch.randint(0, 256, (a.batch, a.height, a.width, 3), generator=g).float() - 110.0).to(dev, dt)
    preds = plan.forward(x)
    torch.cuda.synchronize()
    assert torch.isfinite(preds.float()).all()
    table = plan.layer_table()
    best = None
    for _ in range(5):
        _, ms = plan.forward_timed(x, preds)
        best = ms if best is None else [min(p, q) for p, q in zip(best, ms)]
    print("%-52s %8s %9s %8s" % ("launch", "us", "GB/s", "TF/s"))
    for (name, fl, by), ms in zip(table, best):
        print("%-52s %8.1f %9.0f %8.1f" % (name[-52:], ms * 1e3, by / ms / 1e6, fl / ms / 1e9))
    tot_ms, tot_fl = sum(best), sum(f for _, f, _ in table)
    print("sum of launches: %.3f ms  (%.1f TF/s, %.0f img/s)" % (tot_ms, tot_fl / tot_ms / 1e9, a.batch / tot_ms * 1e3))
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        plan.forward(x, preds)
    st.record()
    for _ in range(a.steps):
        plan.forward(x, preds)
    en.record()
    en.synchronize()
    ms = st.elapsed_time(en) / a.steps
    print("forward: %.3f ms/step  %.0f img/s  %.1f TF/s" % (ms, a.batch / ms * 1e3, tot_fl / ms / 1e9))


if __name__ == "__main__":
    main()
