#!/usr/bin/env python
"""bench.py -- SqueezeDet inference throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the whole hot path over one batch already resident in HBM:
sqdet_net_forward (conv1 .. conv12) -> sqdet_interpret_output -> sqdet_filter_prediction.
Workload at every N: BASELINE.json configs[1], "SqueezeDet fp16 inference batch=32 on 1
MI355X, synthetic 1242x375 images", one batch of 32 per GPU (weak scaling, independent
images: no data-path collective).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 200 steps ~ 0.16 s of GPU time; on shared boxes a single ~40 ms stall inside a 30-step (24 ms) timed
    # region was observed to halve the reported rate, 200 steps bound such a hiccup to ~20 %
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step (configs[1]: 32)")
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="run decode + NMS on the forward's stream (no side stream)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--layer-table", default="", help="write the per-launch table (json) here")
    return ap.parse_args(argv)


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


def build_model(args, device_index):
    import squeezedet_amd as S
    from squeezedet_amd import nets, synthetic
    mc = S.kitti_squeezeDet_config_for_input(args.height, args.width)
    mc.LOAD_PRETRAINED_MODEL = False
    mc.BATCH_SIZE = args.batch
    tdt = torch.float16 if args.dtype == "fp16" else torch.float32
    model = nets.SqueezeDet(mc, gpu_id=str(device_index), dtype=tdt)
    params = synthetic.synthetic_params(model, seed=0)
    model.load_params(params)
    x = synthetic.synthetic_images(args.batch, args.height, args.width, seed=100 + device_index)
    return model, mc, params, x.to(model.device, tdt).contiguous()


def cpu_baseline(args, mc_unused, params, seconds):
    """The oracle (CPU restatement of the reference path: PyTorch-CPU fp32 convs with TF padding
    + NumPy interpret_output + the restated filter_prediction) timed on this box's host cores
    on a bounded sample of the same workload."""
    from oracle import sqdet_oracle as O
    mc = O.squeezeDet_config_for_input(args.height, args.width)
    p32 = O.init_params("squeezeDet", seed=0, storage="fp32")
    nb = 4
    x = O.synthetic_images(nb, args.height, args.width, seed=7)
    O.detect("squeezeDet", mc, p32, x[:1])  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        O.detect("squeezeDet", mc, p32, x)
        n += nb
        if time.perf_counter() - t0 >= seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "images/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "%d synthetic %dx%d images (batches of %d), fp32: PyTorch-CPU convs with TF SAME padding + NumPy "
                      "interpret_output + restated filter_prediction; host has %d logical cores; oneDNN is faster than "
                      "TF-1.0 Eigen, so this over-estimates the reference's own CPU path" % (n, args.width, args.height, nb, os.cpu_count())}


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

    model, mc, params, x = build_model(args, local_rank)
    plan = model._native_plan(args.batch)
    layers = plan.layer_table()

    def step():
        # forward on this stream; interpret_output + filter_prediction of the SAME batch on a side HIP stream behind
        # an event (two-stage pipeline: the next batch's forward overlaps this batch's decode + NMS).  Every step
        # does all of its work inside the timed region: the closing torch.cuda.synchronize() is device-wide.
        if args.no_pipeline:
            boxes, probs, cls = model.detect(x)
            return model.filter_prediction_batch(boxes, probs, cls)
        return model.detect_filter_pipelined(x)

    # ---- warm-up (untimed); the first pass also sizes every buffer ----
    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()

    # untimed per-launch survey to find the dominant kernel (HIP events on the launch stream)
    _, ms0 = plan.forward_timed(x)
    for _ in range(2):
        _, ms = plan.forward_timed(x)
        ms0 = [min(a, b) for a, b in zip(ms0, ms)]
    dom = int(np.argmax(ms0))
    plan.set_probe(dom, args.steps)

    # ---- timed region: EXACTLY `steps` steps between barrier+synchronize pairs ----
    barrier(world, device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    t_issued = time.perf_counter() - t0      # host side done enqueueing (diagnostic: host-bound if ~ elapsed)
    torch.cuda.synchronize()
    barrier(world, device)
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, world, device)

    probe_ms = plan.read_probe(args.steps)
    plan.set_probe(-1, 0)
    counts = out[4].cpu().numpy()
    assert (counts >= 0).all() and (counts <= 64).all()

    if rank == 0:
        value = aggregate_throughput(args.batch, args.steps, world, elapsed)
        name, flops, nbytes = layers[dom]
        avg_ms = float(np.mean(probe_ms)) if probe_ms else float(ms0[dom])
        intensity = flops / nbytes if nbytes else 0.0
        ridge = MFMA_F16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        if args.dtype == "fp16" and intensity > ridge:
            roof = {"bound": "mfma", "achieved": round(flops / (avg_ms * 1e-3) / 1e12, 3), "peak": MFMA_F16_PEAK_TFLOPS,
                    "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": round(nbytes / (avg_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        # HBM bytes per launch from the PMC counters: not measurable inside this process; taken from the
        # committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
        # (profiles/r01_j_hbm_traffic_pmc.json: 2*FETCH_SIZE + WRITE_SIZE, gfx950 correction), else null
        roof["traffic"] = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_j_hbm_traffic_pmc.json")) as fh:
                pmc = json.load(fh)["by_layer"]
            if args.dtype == "fp16" and (args.batch, args.height, args.width) == (32, 375, 1242):
                roof["traffic"] = pmc.get(name)
        except (OSError, KeyError, ValueError):
            pass
        roof["kernel"] = name
        roof["avg_launch_ms"] = round(avg_ms, 5)
        roof["algorithmic_bytes_per_launch"] = nbytes
        roof["algorithmic_flops_per_launch"] = flops
        table = [{"layer": n, "ms": round(m, 5), "GB/s": round(b / (m * 1e-3) / 1e9, 1) if m > 0 else None,
                  "TFLOP/s": round(f / (m * 1e-3) / 1e12, 2) if m > 0 else None, "bytes": b, "flops": f}
                 for (n, f, b), m in zip(layers, ms0)]
        if args.layer_table:
            with open(args.layer_table, "w") as fh:
                json.dump({"layers": table, "forward_ms_sum": sum(ms0), "step_ms": elapsed / args.steps * 1e3}, fh, indent=1)
        res = {
            "metric": "images/sec SqueezeDet 1242x375 inference",
            "value": round(value, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16" if args.dtype == "fp16" else "f32",
            "data": "synthetic",
            "config": {"workload": "SqueezeDet %s inference, batch=%d per GPU, synthetic %dx%d images, full hot path "
                                   "(forward + interpret_output + filter_prediction), inputs resident in HBM"
                                   % (args.dtype, args.batch, args.width, args.height),
                       "global_batch": args.batch * world, "parallelism": "dp%d (independent image shards, no collective)" % world},
            "roofline": roof,
            "host_issue_ms_per_step": round(t_issued / args.steps * 1e3, 4),
            "forward_launches_ms_sum": round(float(sum(ms0)), 4),
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args, mc, params, args.cpu_baseline_seconds)
        print(json.dumps(res), flush=True)
    if _dist_on():
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
