"""Helper of tests/test_gpu_dist.py, run under `python -m torch.distributed.run --nproc-per-node 1` (backend nccl = RCCL):
two SqueezeDetTrainer steps WITHOUT a process group, then the same two steps from the same initial weights WITH the RCCL
group of this launch (world size 1: broadcast, gradient-bucket all-reduce and the scalar num_objects all-reduce all run
through RCCL), replica-mean and global-num_objects normalisation -- the updated variables must be bitwise equal."""
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
    from oracle import train_oracle as TO
    from squeezedet_amd import nets
    from squeezedet_amd.train import SqueezeDetTrainer
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def make(**kw):
        mc = S.kitti_squeezeDet_config_for_input(128, 256)
        mc.LOAD_PRETRAINED_MODEL = False
        mc.IS_TRAINING = True
        mc.BATCH_SIZE = 2
        m = nets.SqueezeDet(mc, gpu_id=str(local_rank), dtype=torch.float32)
        m.load_params(O.init_params("squeezeDet", seed=3))
        return SqueezeDetTrainer(m, seed=11, **kw)

    omc = O.squeezeDet_config_for_input(128, 256)
    x = O.synthetic_images(2, 128, 256, seed=21)
    mask, delta, box, labels = TO.synthetic_labels(omc, 2, seed=22)

    def two_steps(tr):
        for _ in range(2):
            out = tr.step(x, mask, delta, box, labels)
        torch.cuda.synchronize()
        return tr.flat_params.clone(), float(out["class_loss"]) + float(out["conf_loss"]) + float(out["bbox_loss"])

    p_single, l_single = two_steps(make())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    tr = make()
    assert tr.world == world
    p_dist, l_dist = two_steps(tr)
    p_glob, l_glob = two_steps(make(global_num_objects=True, process_group=dist.group.WORLD))
    res = {"world": world, "replica_mean_equal": bool(torch.equal(p_single, p_dist)), "global_equal": bool(torch.equal(p_single, p_glob)),
           "loss_single": l_single, "loss_dist": l_dist, "loss_global": l_glob, "moved": float((p_single - make().flat_params).abs().max())}
    dist.barrier(device_ids=[local_rank])
    dist.destroy_process_group()
    if rank == 0:
        print("DIST_CHECK " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
