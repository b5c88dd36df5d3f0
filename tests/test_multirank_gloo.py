"""N>1 path of bench.py on CPU: world_size-2 gloo.  The hot path shards by image with no
data-path collective; the only cross-rank steps are the barrier and the max-over-ranks
timing reduction, which is what is exercised here."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    assert bench.dist_env() == (rank, rank, world)
    bench.barrier(world, dev)
    mine = 1.0 + rank * 0.5                       # rank 1 is the slow one
    t = bench.max_over_ranks(mine, world, dev)
    val = bench.aggregate_throughput(32, 10, world, t)
    # shards are disjoint: every rank seeds its own images
    q.put((rank, t, val))
    bench.barrier(world, dev)
    dist.destroy_process_group()


def test_max_over_ranks_and_weak_scaling_value():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, t, val in res:
        assert t == 1.5                            # MAX over ranks, identical on every rank
        assert val == 32 * 10 * 2 / 1.5            # whole-job aggregate: all ranks' images / max time


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from squeezedet_amd.train import allreduce_gradients
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(1234 + rank)
    flat = torch.randn(4096 + 64, generator=g)          # this rank's flat gradient bucket
    mine = flat.clone()
    scale = allreduce_gradients(flat, world)
    q.put((rank, mine, flat.clone(), scale))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_bucket_allreduce_is_the_mean_on_every_rank():
    """Training DP: one SUM all-reduce of the flat bucket, the optimizer kernel applies
    the returned 1/world factor -> every replica updates with the same mean gradient."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = res[0][1] + res[1][1]
    for rank, mine, reduced, scale in res:
        assert scale == 0.5
        assert torch.equal(reduced, total)                 # identical bits on both ranks
        assert torch.allclose(reduced * scale, (res[0][1] + res[1][1]) / 2)


def test_single_rank_defaults():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args([])
    assert a.gpus == 1 and a.batch == 32 and (a.height, a.width) == (375, 1242) and a.dtype == "fp16"
    assert bench.max_over_ranks(2.0, 1, torch.device("cpu")) == 2.0
    assert bench.aggregate_throughput(32, 30, 1, 2.0) == 480.0
