"""Helper of tests/test_gpu_dist.py::test_world2_trainer_on_one_gpu, run under `python -m torch.distributed.run --nproc-per-node 2`:
TWO ranks that BOTH use cuda:0 (RCCL refuses two ranks on one device, gloo does not), backend gloo, the real
SqueezeDetTrainer + GraphedStep: rank-0 broadcast of the variables, the eager num_objects all-reduce ahead of the graph replay,
the SUM all-reduce of the flat DEVICE gradient bucket, both loss normalisations.  Every rank first runs the SINGLE-process steps
at the global batch 2B (before the process group exists), then the two-rank steps on its half of the same batch with the same
dropout masks.  Printed by rank 0: DIST2_CHECK {json}."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch.distributed as dist

    import squeezedet_amd as S
    from oracle import sqdet_oracle as O
    from squeezedet_amd import nets, ops
    from squeezedet_amd import train as T
    from tools.bench_train import synthetic_ground_truth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == 2
    torch.cuda.set_device(0)                      # both ranks share the one GPU
    dev = torch.device("cuda", 0)
    B, H, W, STEPS = 2, 128, 256, 2

    def make(seed, **kw):
        mc = S.kitti_squeezeDet_config_for_input(H, W)
        mc.LOAD_PRETRAINED_MODEL = False
        mc.IS_TRAINING = True
        mc.BATCH_SIZE = kw.pop("batch")
        m = nets.SqueezeDet(mc, gpu_id="0", dtype=torch.float32)
        m.load_params(O.init_params("squeezeDet", seed=seed))
        return mc, T.SqueezeDetTrainer(m, seed=11, **kw)

    mc0 = S.kitti_squeezeDet_config_for_input(H, W)
    xg = O.synthetic_images(2 * B, H, W, seed=21).to(dev)
    gt, gcls, gcnt = [torch.from_numpy(a).to(dev) for a in synthetic_ground_truth(mc0, 2 * B, seed=22)]
    anchors = torch.from_numpy(np.asarray(mc0.ANCHOR_BOX, np.float64)).to(dev)

    # one dropout mask per step for the GLOBAL batch; a replica uses its images' rows of it
    state = {"off": 0, "calls": 0}
    def mask_into(mask, keep, seed):
        state["calls"] += 1
        g = ops.dropout_mask((2 * B,) + tuple(mask.shape[1:]), keep, 4242 + state["calls"], mask.dtype, mask.device)
        mask.copy_(g[state["off"]:state["off"] + mask.shape[0]])
    ops.dropout_mask_into = mask_into

    def run_steps(tr, mc, lo, hi):
        state["off"], state["calls"] = lo, 0
        stepper = T.GraphedStep(tr, anchors, mc.CLASSES)
        for _ in range(STEPS):
            out = stepper.step(xg[lo:hi], gt[lo:hi], gcls[lo:hi], gcnt[lo:hi])
        torch.cuda.synchronize()
        tr.flush()
        return tr.flat_params.clone(), [float(out[k]) for k in ("class_loss", "conf_loss", "bbox_loss")]

    # ---- single process, global batch 2B (rank 0's initial weights = seed 3): the reference graph of both modes
    mc, tr = make(3, batch=2 * B)
    p_init = tr.flat_params.clone()
    p_single, _ = run_steps(tr, mc, 0, 2 * B)

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {"world": dist.get_world_size(), "backend": dist.get_backend()}
    for mode in ("global", "replica_mean"):
        # rank 1 starts from OTHER weights: the constructor's broadcast must replace them with rank 0's
        mc, tr = make(3 if rank == 0 else 5, batch=B, global_num_objects=(mode == "global"), process_group=dist.group.WORLD)
        assert tr.world == 2 and tr.rank == rank
        res[mode + "_broadcast_ok"] = bool(torch.equal(tr.flat_params, p_init))
        p, losses = run_steps(tr, mc, rank * B, (rank + 1) * B)
        both = [torch.empty_like(p) for _ in range(2)]
        dist.all_gather(both, p)                                      # (device tensors through gloo)
        res[mode + "_ranks_bitwise_equal"] = bool(torch.equal(both[0], both[1]))
        res[mode + "_moved"] = float((p - p_init).abs().max())
        res[mode + "_max_abs_diff_vs_single_2B"] = float((p - p_single).abs().max())
        res[mode + "_skipped"] = int(tr.skipped_steps)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("DIST2_CHECK " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
