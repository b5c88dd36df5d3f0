#!/usr/bin/env python
"""Training throughput of SqueezeDet on MI355X: BASELINE.json configs[2], "SqueezeDet fp32 training
batch=20/GPU, RCCL grad all-reduce, synthetic KITTI labels" (network input 1248x384, the reference's
training size).  One step = forward (dropout on) + loss + backward + flat-bucket gradient all-reduce
+ clipped Momentum update.  Launch with torch.distributed.run for N > 1 (one process per GPU).

    python tools/bench_train.py --steps 10 --warmup 3
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synthetic_dense_labels(mc, batch, seed):
    """Seeded KITTI-like dense labels without the oracle: n~U{1..8} boxes per image,
    each assigned to the anchor of highest IoU (dataset/imdb.py:195-239 semantics, vectorised)."""
    rs = np.random.RandomState(seed)
    anchor = np.asarray(mc.ANCHOR_BOX)
    A, C = mc.ANCHORS, mc.CLASSES
    mask = np.zeros((batch, A), np.float32)
    delta = np.zeros((batch, A, 4), np.float32)
    box = np.zeros((batch, A, 4), np.float32)
    labels = np.zeros((batch, A, C), np.float32)
    for b in range(batch):
        for _ in range(rs.randint(1, 9)):
            g = np.array([rs.uniform(0, mc.IMAGE_WIDTH), rs.uniform(0, mc.IMAGE_HEIGHT), rs.uniform(20, 300), rs.uniform(20, 200)])
            lr = np.maximum(np.minimum(anchor[:, 0] + anchor[:, 2] / 2, g[0] + g[2] / 2) - np.maximum(anchor[:, 0] - anchor[:, 2] / 2, g[0] - g[2] / 2), 0)
            tb = np.maximum(np.minimum(anchor[:, 1] + anchor[:, 3] / 2, g[1] + g[3] / 2) - np.maximum(anchor[:, 1] - anchor[:, 3] / 2, g[1] - g[3] / 2), 0)
            inter = lr * tb
            iou = inter / (anchor[:, 2] * anchor[:, 3] + g[2] * g[3] - inter)
            iou[mask[b] > 0] = -1
            a = int(np.argmax(iou)) if iou.max() > 0 else int(np.argmin(((anchor - g) ** 2).sum(1) + 1e12 * (mask[b] > 0)))
            mask[b, a] = 1
            delta[b, a] = [(g[0] - anchor[a, 0]) / anchor[a, 2], (g[1] - anchor[a, 1]) / anchor[a, 3], np.log(g[2] / anchor[a, 2]), np.log(g[3] / anchor[a, 3])]
            box[b, a] = g
            labels[b, a, rs.randint(0, C)] = 1
    return mask, delta, box, labels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default: 20 squeezeDet, 8 resnet50)")
    ap.add_argument("--arch", default="squeezeDet", choices=["squeezeDet", "resnet50"],
                    help="squeezeDet = BASELINE.json configs[2]; resnet50 = configs[4] (ResNet50+ConvDet, 1242x375, in float32)")
    args = ap.parse_args()
    if args.batch <= 0:
        args.batch = 20 if args.arch == "squeezeDet" else 8
    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import squeezedet_amd as S
    from squeezedet_amd import nets, synthetic
    from squeezedet_amd.train import ResNet50ConvDetTrainer, SqueezeDetTrainer
    mc = S.kitti_squeezeDet_config() if args.arch == "squeezeDet" else S.kitti_res50_config()
    mc.LOAD_PRETRAINED_MODEL = False
    mc.IS_TRAINING = True
    mc.BATCH_SIZE = args.batch
    cls, trainer = (nets.SqueezeDet, SqueezeDetTrainer) if args.arch == "squeezeDet" else (nets.ResNet50ConvDet, ResNet50ConvDetTrainer)
    model = cls(mc, gpu_id=str(local_rank), dtype=torch.float32)
    model.load_params(synthetic.synthetic_params(model, seed=0))      # same weights on every rank
    tr = trainer(model)
    x = synthetic.synthetic_images(args.batch, mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH, seed=100 + rank).to(dev)
    lab = [torch.from_numpy(a).to(dev) for a in synthetic_dense_labels(mc, args.batch, seed=200 + rank)]
    for _ in range(args.warmup):
        out = tr.step(x, *lab)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = tr.step(x, *lab)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank == 0:
        label = "SqueezeDet 1248x384" if args.arch == "squeezeDet" else "ResNet50+ConvDet 1242x375"
        print(json.dumps({"metric": "images/sec %s fp32 training" % label, "value": round(args.batch * world * args.steps / el, 2),
                          "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(el / args.steps * 1e3, 3), "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "%s fp32 training, batch=%d per GPU, forward+loss+backward+"
                                                 "all-reduce+clipped Momentum" % (label, args.batch), "parallelism": "dp%d" % world},
                          "losses": {k: float(out[k]) for k in ("class_loss", "conf_loss", "bbox_loss")}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
