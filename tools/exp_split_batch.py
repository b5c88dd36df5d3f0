#!/usr/bin/env python
"""Experiment: the batch-32 forward as two batch-16 plans on two HIP streams (tails of one half under the heads of the other)
against the one-plan forward; forward only (no post-processing), rotating inputs.   python tools/exp_split_batch.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import squeezedet_amd as S  # noqa: E402
from squeezedet_amd import nets, ops, synthetic  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
mc = S.kitti_squeezeDet_config_for_input(375, 1242)
mc.BATCH_SIZE = 32
mc.LOAD_PRETRAINED_MODEL = False
model = nets.SqueezeDet(mc, gpu_id="0", dtype=torch.float16)
model.load_params(synthetic.synthetic_params(model, seed=0))
g = torch.Generator(device=dev).manual_seed(0)
xs = [torch.randn(32, 375, 1242, 3, device=dev, generator=g).half() for _ in range(4)]


def mkplan(b):
    p = ops.NetPlan(model.NATIVE_ARCH, model.dtype, b, mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH, mc.CLASSES, mc.ANCHOR_PER_GRID, model.device)
    p.set_bn_epsilon(mc.BATCH_NORM_EPSILON)
    for name, t in model.params.items():
        p.set_param(name, t)
    return p


full = mkplan(32)
halves = [mkplan(16), mkplan(16)]
quarters = [mkplan(8) for _ in range(4)]
preds = torch.empty((32, full.gh, full.gw, full.out_ch), dtype=torch.float16, device=dev)
streams = [torch.cuda.Stream() for _ in range(4)]
ev = [torch.cuda.Event() for _ in range(4)]
main = torch.cuda.current_stream()


def step_full(i):
    full.forward(xs[i & 3], preds)


def step_split(plans):
    n = len(plans)
    b = 32 // n

    def f(i):
        x = xs[i & 3]
        e0 = torch.cuda.Event()
        e0.record(main)
        for k in range(n):
            with torch.cuda.stream(streams[k]):
                streams[k].wait_event(e0)
                plans[k].forward(x[k * b:(k + 1) * b], preds[k * b:(k + 1) * b])
                ev[k].record(streams[k])
        for k in range(n):
            main.wait_event(ev[k])
    return f


def bench(fn, steps=200):
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


ref = preds.clone()
step_full(0); torch.cuda.synchronize(); ref = preds.clone()
step_split(halves)(0); torch.cuda.synchronize()
print("halves bitwise equal to one plan:", bool(torch.equal(ref, preds)))
for rep in range(2):
    print("one plan (batch 32):      %.4f ms" % bench(step_full))
    print("two plans (2 x 16):       %.4f ms" % bench(step_split(halves)))
    print("four plans (4 x 8):       %.4f ms" % bench(step_split(quarters)))
