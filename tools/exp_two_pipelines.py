#!/usr/bin/env python
"""Experiment (round 4): TWO independent batch-32 forwards in flight on two HIP streams (two plans, two workspaces, no cross-stream
events) against one: do the ramps / tails of one stream's launches fill under the other's?  Forward only, rotating inputs.
    python tools/exp_two_pipelines.py"""
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


plans = [mkplan(32), mkplan(32)]
preds = [torch.empty((32, plans[0].gh, plans[0].gw, plans[0].out_ch), dtype=torch.float16, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
K = 100


def run(nstreams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        s = i % nstreams
        with torch.cuda.stream(streams[s]):
            plans[s].forward(xs[i & 3], preds[s])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


for _ in range(2):
    run(1); run(2)
for rep in range(3):
    print("one stream: %.4f ms per forward   two streams: %.4f ms per forward" % (run(1), run(2)), flush=True)
# does the PAIR of streams matter (hardware-queue assignment)?  Later streams of torch's pool, and explicitly prioritised ones
pool = [torch.cuda.Stream() for _ in range(10)]
for a, b in ((0, 1), (2, 3), (4, 5), (6, 7), (8, 9), (0, 5), (1, 8)):
    streams[0], streams[1] = pool[a], pool[b]
    run(2)
    print("pool streams (%d, %d): %.4f ms per forward" % (a, b, run(2)), flush=True)
hp = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)]
streams[0], streams[1] = hp
run(2)
print("two high-priority streams: %.4f ms per forward" % run(2))
streams[0], streams[1] = torch.cuda.current_stream(), pool[0]
run(2)
print("default stream + one pool stream: %.4f ms per forward" % run(2))
