#!/usr/bin/env python
"""Training throughput of SqueezeDet on MI355X: BASELINE.json configs[2], "SqueezeDet fp32 training
batch=20/GPU, RCCL grad all-reduce, synthetic KITTI labels" (network input 1248x384, the reference's
training size).  One step = GPU label build + forward (dropout on) + loss + backward + flat-bucket gradient all-reduce
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


def synthetic_ground_truth(mc, batch, seed, max_objects=8):
    """Seeded KITTI-like ground truth: n~U{1..8} boxes per image, w in [20,300], h in [20,200],
    centre uniform in the image, class U{0..C-1} -- as padded arrays for sqdet_build_labels."""
    rs = np.random.RandomState(seed)
    gt = np.zeros((batch, max_objects, 4), np.float64)
    cls = rs.randint(0, mc.CLASSES, size=(batch, max_objects)).astype(np.int32)
    cnt = rs.randint(1, max_objects + 1, size=batch).astype(np.int32)
    gt[..., 0] = rs.uniform(0, mc.IMAGE_WIDTH, (batch, max_objects))
    gt[..., 1] = rs.uniform(0, mc.IMAGE_HEIGHT, (batch, max_objects))
    gt[..., 2] = rs.uniform(20, 300, (batch, max_objects))
    gt[..., 3] = rs.uniform(20, 200, (batch, max_objects))
    return gt, cls, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default: 20 squeezeDet, 8 resnet50)")
    ap.add_argument("--arch", default="squeezeDet", choices=["squeezeDet", "resnet50"],
                    help="squeezeDet = BASELINE.json configs[2]; resnet50 = configs[4] (ResNet50+ConvDet, 1242x375, in float32)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"],
                    help="f32 = the reference's training dtype; f16 = mixed precision (float16 activations / activation gradients, "
                         "float32 master weights, weight gradients and optimizer, dynamic loss scale) -- configs[4] is resnet50 + f16")
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
    model = cls(mc, gpu_id=str(local_rank), dtype=torch.float32 if args.dtype == "f32" else torch.float16)
    model.load_params(synthetic.synthetic_params(model, seed=0))      # same weights on every rank
    tr = trainer(model, lazy_overflow_check=True)
    x = synthetic.synthetic_images(args.batch, mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH, seed=100 + rank).to(dev)
    # ground truth lives on the device; the anchor assignment + dense label build (imdb.py:195-239,
    # train.py:163-224) runs on the GPU inside every step, like the reference's per-batch _load_data
    from squeezedet_amd import ops
    anchors = torch.from_numpy(np.asarray(mc.ANCHOR_BOX, np.float64)).to(dev)
    gt, gcls, gcnt = [torch.from_numpy(a).to(dev) for a in synthetic_ground_truth(mc, args.batch, seed=200 + rank)]
    nobj = float(gcnt.sum().item())      # known to the host: no per-step device -> host sync for sum(input_mask)
    one_step = lambda: tr.step(x, *ops.build_labels(anchors, gt, gcls, gcnt, mc.CLASSES)[:4], num_objects=nobj)
    for _ in range(args.warmup):
        out = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    t_issued = time.perf_counter() - t0          # host done enqueueing (diagnostic: launch-bound if ~ the step time)
    torch.cuda.synchronize()
    tr.flush()
    if world > 1:
        dist.barrier(device_ids=[local_rank])
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank == 0:
        label = "SqueezeDet 1248x384" if args.arch == "squeezeDet" else "ResNet50+ConvDet 1242x375"
        prec = "fp32" if args.dtype == "f32" else "fp16 (mixed precision)"
        print(json.dumps({"metric": "images/sec %s %s training" % (label, prec), "value": round(args.batch * world * args.steps / el, 2),
                          "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(el / args.steps * 1e3, 3),
                          "host_issue_ms_per_step": round(t_issued / args.steps * 1e3, 3), "dtype": args.dtype, "data": "synthetic",
                          "config": {"workload": "%s %s training, batch=%d per GPU, forward+loss+backward+"
                                                 "all-reduce+clipped Momentum" % (label, prec, args.batch), "parallelism": "dp%d" % world},
                          "skipped_steps": tr.skipped_steps, "loss_scale": tr.loss_scale,
                          "losses": {k: float(out[k]) for k in ("class_loss", "conf_loss", "bbox_loss")}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
