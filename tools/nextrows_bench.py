#!/usr/bin/env python
"""Measurement of the SURVEY.md section 8(f) "next" rows either side of the hot path, and of the serving loop when the
boundary hands over HOST images (the PCIe-inclusive rate DESIGN.md section 5 quotes; never bench.py's `value`):

  N1  sqdet_preprocess_bgr   uint8 BGR camera frames -> resized, mean-subtracted fp16 network input   (demo.py:186-190)
  N2  sqdet_build_labels     ground-truth boxes -> anchor assignment + dense label tensors             (imdb.py:195-239)
  host-fed serving           pinned uint8 frames --H2D (copy stream)--> preprocess -> forward -> decode + filter -> pinned rows

    python tools/nextrows_bench.py [--batch 32] [--steps 100]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import squeezedet_amd as S  # noqa: E402
from squeezedet_amd import nets, ops, synthetic  # noqa: E402
from tools.kbench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=100)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    mc = S.kitti_squeezeDet_config_for_input(375, 1242)
    mc.BATCH_SIZE = a.batch
    mc.LOAD_PRETRAINED_MODEL = False
    out = {}

    # ---- N1: camera frames are KITTI-sized (375 x 1242) uint8 BGR; the config's input is the same size (identity resize)
    # and 384 x 1248 (a real resize)
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (a.batch, 375, 1242, 3), generator=g, dtype=torch.uint8).to(dev)
    for (dh, dw) in ((375, 1242), (384, 1248)):
        ms = timeit(lambda: ops.preprocess_bgr(frames, dh, dw, mc.BGR_MEANS, torch.float16), 50)
        nbytes = frames.numel() + a.batch * dh * dw * 3 * 2
        out["N1 preprocess_bgr %dx%d -> %dx%d fp16" % (1242, 375, dw, dh)] = {
            "us": round(ms * 1e3, 1), "algorithmic_MB": round(nbytes / 1e6, 1), "GB/s": round(nbytes / ms / 1e6, 0),
            "frac_of_8TB/s": round(nbytes / ms / 1e6 / 8000.0, 3), "images/s": round(a.batch / ms * 1e3)}

    # ---- N2: 20 images x up to 32 objects (train.py's batch), 16848 anchors, float64 IoU
    mct = S.kitti_squeezeDet_config()
    rs = np.random.RandomState(1)
    B, M = 20, 32
    gt = np.zeros((B, M, 4))
    gt[..., 0] = rs.uniform(50, 1200, (B, M)); gt[..., 1] = rs.uniform(30, 350, (B, M))
    gt[..., 2] = rs.uniform(20, 300, (B, M)); gt[..., 3] = rs.uniform(20, 200, (B, M))
    gcls = rs.randint(0, 3, (B, M)); gcnt = rs.randint(1, M + 1, B)
    gtd, gcd, gnd = torch.from_numpy(gt).to(dev), torch.from_numpy(gcls).to(dev, torch.int32), torch.from_numpy(gcnt).to(dev, torch.int32)
    anc = torch.from_numpy(np.asarray(mct.ANCHOR_BOX)).to(dev)
    ms = timeit(lambda: ops.build_labels(anc, gtd, gcd, gnd, mct.CLASSES), 50)
    A = anc.shape[0]
    wbytes = B * A * (1 + 4 + 4 + mct.CLASSES) * 4          # the four dense tensors it writes
    out["N2 build_labels batch 20 x <=32 objects x %d anchors" % A] = {
        "us": round(ms * 1e3, 1), "written_MB": round(wbytes / 1e6, 1), "GB/s": round(wbytes / ms / 1e6, 0),
        "boxes/s": round(float(gcnt.sum()) / ms * 1e3)}

    # ---- host-fed serving loop: two pinned frame buffers, H2D on a copy stream, everything else as bench.py's step
    model = nets.SqueezeDet(mc, gpu_id="0", dtype=torch.float16)
    model.load_params(synthetic.synthetic_params(model, seed=0))
    host = [torch.randint(0, 256, (a.batch, 375, 1242, 3), generator=g, dtype=torch.uint8).pin_memory() for _ in range(2)]
    dbuf = [torch.empty((a.batch, 375, 1242, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    copy_stream = torch.cuda.Stream()
    copied = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]
    main_stream = torch.cuda.current_stream()

    def step(i):
        s = i & 1
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[s])                  # the frames of step i-2 have been preprocessed
            dbuf[s].copy_(host[s], non_blocking=True)
            copied[s].record(copy_stream)
        main_stream.wait_event(copied[s])
        x = ops.preprocess_bgr(dbuf[s], mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH, mc.BGR_MEANS, torch.float16)
        consumed[s].record(main_stream)
        return model.detect_filter_pipelined(x, to_host=True)

    for s in range(2):
        consumed[s].record(main_stream)
    for i in range(10):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    h2d = a.batch * 375 * 1242 * 3
    out["host-fed serving (pinned uint8 frames -> pinned filtered rows), batch %d" % a.batch] = {
        "ms_per_step": round(el / a.steps * 1e3, 4), "images/s": round(a.batch * a.steps / el),
        "H2D_GB/s": round(h2d * a.steps / el / 1e9, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
