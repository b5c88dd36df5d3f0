#!/usr/bin/env python
"""bench.py -- the SqueezeDet hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--config (default sqdet_infer = BASELINE.json configs[1], the headline):
  sqdet_infer        SqueezeDet float16 inference, batch 32 per GPU, synthetic 1242x375 images.  One step = one pass of
                     the whole hot path over one batch already resident in HBM: sqdet_net_forward (conv1 .. conv12) ->
                     sqdet_interpret_output -> sqdet_filter_prediction -> the <= 64 filtered rows per image copied to
                     pinned host memory (what the reference's sess.run + filter_prediction hand the caller).
  sqdetplus_infer    configs[3]: SqueezeDet+ float16 inference, batch 8 per GPU (64 over 8 GPUs), same step.
  sqdet_train_fp32   configs[2]: SqueezeDet float32 training, batch 20 per GPU, 1248x384: GPU label build + forward
                     (dropout on) + loss + backward + flat-bucket gradient all-reduce (RCCL) + clipped Momentum update.
  res50_train_fp16   configs[4]: ResNet50+ConvDet mixed-precision (float16 activations) training, batch 8 per GPU.
Inference shards by image (weak scaling, no data-path collective); training all-reduces one float32 gradient bucket.
Every step reads a DIFFERENT input batch from a rotation larger than the 256 MiB Infinity Cache.
Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak
MFMA_F32_PEAK_TFLOPS = 157.3   # f32-input MFMA (= the f32 vector rate)
MALL_BYTES = 256 << 20

CONFIGS = {
    "sqdet_infer": dict(kind="infer", arch="squeezeDet", batch=32, height=375, width=1242, dtype="fp16",
                        metric="images/sec SqueezeDet 1242x375 inference", steps=200, warmup=20),
    "sqdetplus_infer": dict(kind="infer", arch="squeezeDet+", batch=8, height=375, width=1242, dtype="fp16",
                            metric="images/sec SqueezeDet+ 1242x375 inference", steps=100, warmup=10),
    "sqdet_train_fp32": dict(kind="train", arch="squeezeDet", batch=20, height=384, width=1248, dtype="fp32",
                             metric="images/sec SqueezeDet 1248x384 fp32 training", steps=20, warmup=3),
    "res50_train_fp16": dict(kind="train", arch="resnet50", batch=8, height=375, width=1242, dtype="fp16",
                             metric="images/sec ResNet50+ConvDet 1242x375 fp16 (mixed precision) training", steps=20, warmup=3),
    # (not a BASELINE.json config: configs[2] in mixed precision, for the fp32 / fp16 comparison DESIGN.md quotes)
    "sqdet_train_fp16": dict(kind="train", arch="squeezeDet", batch=20, height=384, width=1248, dtype="fp16",
                             metric="images/sec SqueezeDet 1248x384 fp16 (mixed precision) training", steps=20, warmup=3),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", default="sqdet_infer", choices=sorted(CONFIGS))
    # defaults per config (sqdet_infer: 200 steps ~ 0.14 s of GPU time; short runs reproduce it within a few %)
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--warmup", type=int, default=-1)
    ap.add_argument("--spinup-ms", type=float, default=250.0, help="untimed steps for this long before the timed region (clock settle)")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default: the config's)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--dtype", default="", choices=["", "fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="run decode + NMS on the forward's stream (no side stream)")
    ap.add_argument("--no-graph", action="store_true", help="training configs: issue the step from Python instead of replaying a hipGraph")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--layer-table", default="", help="write the per-launch table (json) here")
    a = ap.parse_args(argv)
    c = CONFIGS[a.config]
    a.kind, a.arch, a.metric = c["kind"], c["arch"], c["metric"]
    a.steps = a.steps or c["steps"]
    a.warmup = a.warmup if a.warmup >= 0 else c["warmup"]
    a.batch = a.batch or c["batch"]
    a.height = a.height or c["height"]
    a.width = a.width or c["width"]
    a.dtype = a.dtype or c["dtype"]
    return a


def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def _dist_on():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def max_over_ranks(value, world, device):
    """Timing rule of the bench contract: the job's time is the MAX over ranks."""
    if not _dist_on():
        return value
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(world, device):
    if _dist_on():
        import torch.distributed as dist
        if device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def aggregate_throughput(images_per_rank_step, steps, world, seconds_max):
    """value = the units ALL ranks processed / the max-over-ranks time."""
    return images_per_rank_step * steps * world / seconds_max


def build_fingerprint():
    """sha256 over the kernel sources the loaded library was built from: a PMC traffic profile is only attached to
    the roofline object when it was taken on THIS build (profiles/*_hbm_traffic_pmc.json carries the fingerprint)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "squeezedet_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".cpp", ".h")):
            with open(os.path.join(csrc, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(config_name, layer):
    """HBM (fabric) bytes per launch of `layer` from the newest committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    of this command whose build fingerprint matches the loaded kernels; None otherwise (stale profiles are refused)."""
    prof = os.path.join(ROOT, "profiles")
    fp = build_fingerprint()
    best = None
    for f in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        if not f.endswith("_hbm_traffic_pmc.json"):
            continue
        try:
            with open(os.path.join(prof, f)) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        if d.get("build_fingerprint") == fp and d.get("config", "sqdet_infer") == config_name and layer in d.get("by_layer", {}):
            best = (d["by_layer"][layer], f)
    return best


def rotation_count(batch_bytes):
    """distinct input batches to rotate through: at least 4, and more than the Infinity Cache holds"""
    return max(4, int(np.ceil(1.3 * MALL_BYTES / float(batch_bytes))))


# ------------------------------------------------------------------------------------------------ inference
def build_infer_model(args, device_index):
    import squeezedet_amd as S
    from squeezedet_amd import nets, synthetic
    if args.arch == "squeezeDet":
        mc = S.kitti_squeezeDet_config_for_input(args.height, args.width)
        cls = nets.SqueezeDet
    else:
        mc = S.kitti_squeezeDetPlus_config()
        assert (args.height, args.width) == (mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH)
        cls = nets.SqueezeDetPlus
    mc.LOAD_PRETRAINED_MODEL = False
    mc.BATCH_SIZE = args.batch
    tdt = torch.float16 if args.dtype == "fp16" else torch.float32
    model = cls(mc, gpu_id=str(device_index), dtype=tdt)
    model.load_params(synthetic.synthetic_params(model, seed=0))
    esz = 2 if args.dtype == "fp16" else 4
    nrot = rotation_count(args.batch * args.height * args.width * 3 * esz)
    xs = [synthetic.synthetic_images(args.batch, args.height, args.width, seed=1000 * device_index + 100 + k).to(model.device, tdt).contiguous()
          for k in range(nrot)]
    return model, mc, xs


def cpu_baseline_infer(args, seconds):
    """The oracle (CPU restatement of the reference path: PyTorch-CPU fp32 convs with TF padding + NumPy
    interpret_output + the restated filter_prediction) timed on this box's host cores on a bounded sample."""
    from oracle import sqdet_oracle as O
    if args.arch == "squeezeDet":
        mc = O.squeezeDet_config_for_input(args.height, args.width)
    else:
        mc = O.kitti_squeezeDetPlus_config()
    p32 = O.init_params(args.arch, seed=0, storage="fp32")
    nb = 4
    x = O.synthetic_images(nb, args.height, args.width, seed=7)
    O.detect(args.arch, mc, p32, x[:1])  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        O.detect(args.arch, mc, p32, x)
        n += nb
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "images/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "%d synthetic %dx%d images (batches of %d), fp32: PyTorch-CPU convs with TF SAME padding + NumPy "
                      "interpret_output + restated filter_prediction; host has %d logical cores; oneDNN is faster than "
                      "TF-1.0 Eigen, so this over-estimates the reference's own CPU path" % (n, args.width, args.height, nb, os.cpu_count())}


def spin_up(step, ms):
    """Untimed steps for a fixed wall time right before the timed region: the W warm-up steps of a short run (the
    driver's 5) end ~3 ms after the per-launch survey's synchronisations, before the clocks have settled -- 20-step runs
    read 8 % below 200-step runs of the same binary without it.  Setup, not measurement: the timed region is still
    exactly K steps between barrier + synchronize pairs."""
    if ms <= 0:
        return
    t0 = time.perf_counter()
    i = 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(8):
            step(i)
            i += 1
        torch.cuda.synchronize()


def launch_roofline(flops, nbytes, ms, dtype):
    """Roofline entry of one launch: MFMA-bound when its algorithmic intensity exceeds the ridge (fp16 only: the peak the
    guide gives is the dense fp16 one), else HBM-bound; achieved = algorithmic flops (bytes) / measured duration."""
    intensity = flops / nbytes if nbytes else 0.0
    ridge = MFMA_F16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    if dtype == "fp16" and intensity > ridge:
        roof = {"bound": "mfma", "achieved": round(flops / (ms * 1e-3) / 1e12, 3), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s"}
    else:
        roof = {"bound": "hbm", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    return roof


def run_infer(args, rank, local_rank, world, device):
    model, mc, xs = build_infer_model(args, local_rank)
    plan = model._native_plan(args.batch)
    layers = plan.layer_table()
    nrot = len(xs)

    def step(i):
        # forward on this stream; interpret_output + filter_prediction + the copy of the filtered rows to pinned host
        # memory on a side HIP stream behind an event (two-stage pipeline: the next batch's forward overlaps this batch's
        # decode + NMS + D2H).  Every step does all of its work inside the timed region: the closing
        # torch.cuda.synchronize() is device-wide.
        x = xs[i % nrot]
        if args.no_pipeline:
            boxes, probs, cls = model.detect(x)
            out = model.filter_prediction_batch(boxes, probs, cls)
            return [t.cpu() for t in out]
        return model.detect_filter_pipelined(x, to_host=True)

    for i in range(max(args.warmup, 1)):
        out = step(i)
    torch.cuda.synchronize()

    # untimed per-launch survey to find the dominant kernel (HIP events on the launch stream)
    _, ms0 = plan.forward_timed(xs[0])
    for k in range(1, 3):
        _, ms = plan.forward_timed(xs[k % nrot])
        ms0 = [min(a, b) for a, b in zip(ms0, ms)]
    dom = int(np.argmax(ms0))
    spin_up(step, args.spinup_ms)
    plan.set_probe(dom, args.steps)

    # ---- timed region: EXACTLY `steps` steps between barrier+synchronize pairs ----
    barrier(world, device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    t_issued = time.perf_counter() - t0      # host side done enqueueing (diagnostic: host-bound if ~ elapsed)
    torch.cuda.synchronize()
    barrier(world, device)
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, world, device)

    probe_ms = plan.read_probe(args.steps)
    plan.set_probe(-1, 0)
    counts = np.asarray(out[4].cpu() if isinstance(out[4], torch.Tensor) else out[4])
    assert (counts >= 0).all() and (counts <= 64).all()
    if rank != 0:
        return None
    value = aggregate_throughput(args.batch, args.steps, world, elapsed)
    name, flops, nbytes = layers[dom]
    avg_ms = float(np.mean(probe_ms)) if probe_ms else float(ms0[dom])
    roof = launch_roofline(flops, nbytes, avg_ms, args.dtype)
    # HBM bytes per launch from the PMC counters: not measurable inside this process; taken from the committed
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command ON THIS BUILD (fingerprint-checked), else null
    tr = pmc_traffic(args.config, name) if (args.batch, args.height, args.width) == tuple(CONFIGS[args.config][k] for k in ("batch", "height", "width")) else None
    roof["traffic"] = tr[0] if tr else None
    if tr:
        roof["traffic_profile"] = tr[1]
    roof["kernel"] = name
    roof["avg_launch_ms"] = round(avg_ms, 5)
    roof["algorithmic_bytes_per_launch"] = nbytes
    roof["algorithmic_flops_per_launch"] = flops
    # the next largest launches (several are within a microsecond of each other, so which one is "dominant" can change from
    # run to run): durations from the untimed survey = one forward at a time, nothing else on the chip
    order = [int(k) for k in np.argsort(ms0)[::-1] if int(k) != dom][:3]
    roof["next_largest_launches"] = [dict(launch_roofline(layers[k][1], layers[k][2], float(ms0[k]), args.dtype), kernel=layers[k][0],
                                          survey_launch_ms=round(float(ms0[k]), 5)) for k in order]
    table = [{"layer": n, "ms": round(m, 5), "GB/s": round(b / (m * 1e-3) / 1e9, 1) if m > 0 else None,
              "TFLOP/s": round(f / (m * 1e-3) / 1e12, 2) if m > 0 else None, "bytes": b, "flops": f}
             for (n, f, b), m in zip(layers, ms0)]
    if args.layer_table:
        with open(args.layer_table, "w") as fh:
            json.dump({"layers": table, "forward_ms_sum": sum(ms0), "step_ms": elapsed / args.steps * 1e3,
                       "build_fingerprint": build_fingerprint()}, fh, indent=1)
    res = result_head(args, value, world, elapsed)
    res["config"] = {"workload": "%s %s inference, batch=%d per GPU, synthetic %dx%d images (%d distinct batches in rotation), full hot "
                                 "path (forward + interpret_output + filter_prediction + filtered rows to pinned host memory), inputs "
                                 "resident in HBM" % (args.arch, args.dtype, args.batch, args.width, args.height, nrot),
                     "name": args.config, "global_batch": args.batch * world,
                     "parallelism": "dp%d (independent image shards, no collective)" % world}
    res["roofline"] = roof
    res["host_issue_ms_per_step"] = round(t_issued / args.steps * 1e3, 4)
    res["forward_launches_ms_sum"] = round(float(sum(ms0)), 4)
    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline_infer(args, args.cpu_baseline_seconds)
    return res


# ------------------------------------------------------------------------------------------------ training
def training_flops_per_image(model):
    """2*MAC of the step: every conv forward; backward-filter of every trainable conv; backward-data of every trainable
    conv except the lowest one (nothing below it needs a gradient)."""
    fwd = bwd = 0.0
    convs = []
    seen = set()

    def walk(n):
        if n in seen:
            return
        seen.add(n)
        for i in n.inputs:
            walk(i)
        if n.op in ("conv", "conv_bn"):
            convs.append(n)
    walk(model.preds)
    first = True
    for n in convs:
        cin = int(n.inputs[0].get_shape()[3])
        _, ho, wo, cout = n.get_shape()
        f = 2.0 * n.attrs["size"] ** 2 * cin * cout * ho * wo
        fwd += f
        if model.trainable[n.name + "/kernels"]:
            bwd += f if first else 2.0 * f
            first = False
    return fwd, bwd


def cpu_baseline_train(args, seconds):
    """The training oracle (PyTorch-CPU float32 autograd restatement of forward + loss + backward) on a bounded sample."""
    from oracle import sqdet_oracle as O
    from oracle import train_oracle as TO
    nb = 2
    if args.arch == "squeezeDet":
        mc = O.squeezeDet_config_for_input(args.height, args.width)
        params = O.init_params("squeezeDet", seed=0)
        x = O.synthetic_images(nb, args.height, args.width, seed=7)
        mask, delta, box, labels = TO.synthetic_labels(mc, nb, seed=3)
        dm = torch.ones((nb, 24, 78, 768))
        fn = lambda: TO.loss_and_grads("squeezeDet", mc, params, x, dm, mask, delta, box, labels)
    else:
        from oracle import resnet_oracle as RO
        mc = O.kitti_res50_config()
        params = RO.init_params(seed=0)
        x = O.synthetic_images(nb, args.height, args.width, seed=7)
        mask, delta, box, labels = TO.synthetic_labels(mc, nb, seed=3)
        dm = torch.ones((nb, 24, 78, 1024))
        fn = lambda: RO.loss_and_grads(mc, params, x, dm, mask, delta, box, labels)
    fn()
    t0 = time.perf_counter()
    n = 0
    while True:
        fn()
        n += nb
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "images/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "%d synthetic %dx%d images (batches of %d): PyTorch-CPU float32 autograd restatement of forward + loss + "
                      "backward (no optimizer step); host has %d logical cores" % (n, args.width, args.height, nb, os.cpu_count())}


def run_train(args, rank, local_rank, world, device):
    import squeezedet_amd as S
    from squeezedet_amd import nets, ops, synthetic
    from squeezedet_amd.train import GraphedStep, ResNet50ConvDetTrainer, SqueezeDetTrainer
    from tools.bench_train import synthetic_ground_truth
    mc = S.kitti_squeezeDet_config() if args.arch == "squeezeDet" else S.kitti_res50_config()
    assert (args.height, args.width) == (mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH), "training configs run at the config's own input size"
    mc.LOAD_PRETRAINED_MODEL = False
    mc.IS_TRAINING = True
    mc.BATCH_SIZE = args.batch
    cls, trainer = (nets.SqueezeDet, SqueezeDetTrainer) if args.arch == "squeezeDet" else (nets.ResNet50ConvDet, ResNet50ConvDetTrainer)
    tdt = torch.float32 if args.dtype == "fp32" else torch.float16
    model = cls(mc, gpu_id=str(local_rank), dtype=tdt)
    model.load_params(synthetic.synthetic_params(model, seed=0))      # same weights on every rank
    tr = trainer(model, lazy_overflow_check=True)
    nrot = 4
    xs = [synthetic.synthetic_images(args.batch, mc.IMAGE_HEIGHT, mc.IMAGE_WIDTH, seed=100 + 10 * rank + k).to(device) for k in range(nrot)]
    anchors = torch.from_numpy(np.asarray(mc.ANCHOR_BOX, np.float64)).to(device)
    gts = [[torch.from_numpy(a).to(device) for a in synthetic_ground_truth(mc, args.batch, seed=200 + 10 * rank + k)] for k in range(nrot)]
    use_graph = not args.no_graph
    stepper = GraphedStep(tr, anchors, mc.CLASSES) if use_graph else None

    def step(i):
        x, (gt, gcls, gcnt) = xs[i % nrot], gts[i % nrot]
        if stepper is not None:
            return stepper.step(x, gt, gcls, gcnt)
        return tr.step(x, *ops.build_labels(anchors, gt, gcls, gcnt, mc.CLASSES)[:4])

    for i in range(max(args.warmup, 1)):
        out = step(i)
    torch.cuda.synchronize()
    spin_up(step, args.spinup_ms)
    barrier(world, device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    t_issued = time.perf_counter() - t0
    torch.cuda.synchronize()
    tr.flush()
    barrier(world, device)
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, world, device)
    if rank != 0:
        return None
    value = aggregate_throughput(args.batch, args.steps, world, elapsed)
    fwd, bwd = training_flops_per_image(model)
    step_ms = elapsed / args.steps * 1e3
    peak = MFMA_F32_PEAK_TFLOPS if args.dtype == "fp32" else MFMA_F16_PEAK_TFLOPS
    flops = (fwd + bwd) * args.batch
    roof = {"bound": "mfma", "achieved": round(flops / (step_ms * 1e-3) / 1e12, 3), "peak": peak, "unit": "TFLOP/s"}
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    roof["traffic"] = None
    roof["kernel"] = "whole training step (forward + loss + backward + update; ~300 launches%s)" % (", replayed as one hipGraph" if use_graph else "")
    roof["avg_launch_ms"] = round(step_ms, 4)
    roof["algorithmic_flops_per_launch"] = flops
    res = result_head(args, value, world, elapsed)
    res["config"] = {"workload": "%s %s training, batch=%d per GPU, synthetic %dx%d images + KITTI-like ground truth (%d distinct batches "
                                 "in rotation): GPU label build + forward + loss + backward + gradient all-reduce + clipped Momentum"
                                 % (args.arch, "float32" if args.dtype == "fp32" else "mixed-precision (float16 activations)", args.batch,
                                    args.width, args.height, nrot),
                     "name": args.config, "global_batch": args.batch * world,
                     "parallelism": "dp%d (one flat float32 gradient bucket all-reduced over RCCL per step)" % world}
    res["roofline"] = roof
    res["host_issue_ms_per_step"] = round(t_issued / args.steps * 1e3, 4)
    res["hipgraph"] = bool(use_graph)
    res["skipped_steps"] = tr.skipped_steps
    res["losses"] = {k: float(out[k]) for k in ("class_loss", "conf_loss", "bbox_loss")}
    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline_train(args, args.cpu_baseline_seconds)
    return res


def result_head(args, value, world, elapsed):
    return {
        "metric": args.metric,
        "value": round(value, 2),
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "spinup_ms": args.spinup_ms,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16" if args.dtype == "fp16" else "f32",
        "data": "synthetic",
    }


def main(argv=None):
    args = parse_args(argv)
    rank, local_rank, world = dist_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world == 1 and args.gpus > 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # (SQDET_FORCE_DIST=1 under torch.distributed.run with one process: exercises the RCCL path on a single-GPU box)
    if world > 1 or (os.environ.get("SQDET_FORCE_DIST") == "1" and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    res = run_infer(args, rank, local_rank, world, device) if args.kind == "infer" else run_train(args, rank, local_rank, world, device)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if _dist_on():
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
