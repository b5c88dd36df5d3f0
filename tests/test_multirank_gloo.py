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


def _trainer_step_worker(rank, world, port, q, global_mode):
    """One data-parallel 'trainer step' on CPU tensors, through the trainer's OWN host logic (squeezedet_amd.train:
    reduce_num_objects, step_normalisation, allreduce_gradients) with the oracle's loss graph standing in for the HIP
    loss kernel: a one-parameter-vector model preds = base * w, each rank holding half of a batch of 4."""
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import sqdet_oracle as O
    from oracle import train_oracle as TO
    from squeezedet_amd.train import allreduce_gradients, reduce_num_objects, step_normalisation
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    mc = O.squeezeDet_config_for_input(64, 128)
    gh, gw = O.squeezedet_grid(64, 128)
    Bl = 2
    rs = np.random.RandomState(77)
    base = torch.from_numpy((rs.randn(Bl * world, gh, gw, 72)).astype(np.float32))
    mask, delta, box, labels = TO.synthetic_labels(mc, Bl * world, seed=78)
    w = torch.full((72,), 1.1, requires_grad=True)
    sl = slice(rank * Bl, (rank + 1) * Bl)
    nobj = torch.as_tensor(mask[sl]).float().sum().reshape(1)
    gb, _ = step_normalisation(global_mode, Bl, world)
    if global_mode:
        reduce_num_objects(nobj, world)                       # the scalar collective
    parts = TO.loss_graph(mc, base[sl] * w, mask[sl], delta[sl], box[sl], labels[sl], num_objects=nobj[0] if global_mode else None,
                          global_batch=gb or None)
    (g,) = torch.autograd.grad(parts["class_loss"] + parts["conf_loss"] + parts["bbox_loss"], w)
    bucket = g.clone()
    allreduce_gradients(bucket, world)                         # the bucket collective (SUM)
    _, scale = step_normalisation(global_mode, Bl, world)
    q.put((rank, float(nobj[0]), (bucket * scale).numpy(), g.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run_trainer_step(global_mode):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_step_worker, args=(r, world, port, q, global_mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _single_process_reference(global_mode):
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import sqdet_oracle as O
    from oracle import train_oracle as TO
    mc = O.squeezeDet_config_for_input(64, 128)
    gh, gw = O.squeezedet_grid(64, 128)
    rs = np.random.RandomState(77)
    base = torch.from_numpy((rs.randn(4, gh, gw, 72)).astype(np.float32))
    mask, delta, box, labels = TO.synthetic_labels(mc, 4, seed=78)

    def grad(sl):
        w = torch.full((72,), 1.1, requires_grad=True)
        parts = TO.loss_graph(mc, base[sl] * w, mask[sl], delta[sl], box[sl], labels[sl])
        return torch.autograd.grad(parts["class_loss"] + parts["conf_loss"] + parts["bbox_loss"], w)[0].numpy()
    if global_mode:
        return grad(slice(0, 4)), float(torch.as_tensor(mask).sum())     # the reference graph at batch 4 (nn_skeleton.py:285-327)
    return (grad(slice(0, 2)) + grad(slice(2, 4))) / 2, None              # mean of two reference graphs at batch 2


def test_world2_trainer_step_global_num_objects_equals_the_full_batch_graph():
    """global_num_objects=True at world 2: num_objects all-reduce + global batch divisor + SUMMED bucket = the gradient of
    the reference's single graph at batch world*B (round 2's bug: the confidence term came out world x too large)."""
    import numpy as np
    res = _run_trainer_step(True)
    ref, nobj = _single_process_reference(True)
    for rank, n, applied, local in res:
        assert n == nobj
        np.testing.assert_allclose(applied, ref, rtol=2e-5, atol=1e-7)
    assert np.array_equal(res[0][2], res[1][2])                          # identical bits on both ranks
    assert not np.allclose(res[0][3], res[1][3])                         # the ranks really held different shards


def test_world2_trainer_step_replica_mean_is_the_mean_of_two_reference_graphs():
    import numpy as np
    res = _run_trainer_step(False)
    ref, _ = _single_process_reference(False)
    for rank, n, applied, local in res:
        np.testing.assert_allclose(applied, ref, rtol=2e-5, atol=1e-7)
    assert np.array_equal(res[0][2], res[1][2])


def _bench(args, timeout=300):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    return r.returncode, json.loads(lines[0])


def test_bench_self_launches_n_ranks_dry_run():
    """`python bench.py --gpus 2` with no torchrun environment (how a driver runs --gpus 1): the script re-launches itself
    under torch.distributed.run; --dry-run skips the kernels so this runs on the CPU box.  ONE JSON line, ranks_seen = 2,
    two distinct processes."""
    rc, res = _bench(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert rc == 0 and res["dry_run"] is True and res["value"] is None
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["config"]["global_batch"] == 64
    assert sorted(d["rank"] for d in res["devices"]) == [0, 1] and len({d["pid"] for d in res["devices"]}) == 2
    # per-rank step times beside the max-reduce: rank 1 sleeps longer than rank 0 in the dry run, and the line's time is the maximum
    rk = res["rank_ms_per_step"]
    assert len(rk["per_rank"]) == 2 and rk["min"] == min(rk["per_rank"]) and rk["max"] == max(rk["per_rank"]) and rk["per_rank"][1] > rk["per_rank"][0]


def test_bench_self_launches_eight_ranks_dry_run():
    """The driver's largest launch -- `--gpus 8`, one process per GPU of a node -- through the same self-launch path: eight distinct
    processes rendezvous on 127.0.0.1, every rank is seen, global batch = 8 x the per-GPU batch (weak scaling), the per-rank times
    arrive in rank order and the maximum is the slowest rank's."""
    rc, res = _bench(["--gpus", "8", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert rc == 0 and res["dry_run"] is True and res["n_gpus"] == 8 and res["ranks_seen"] == 8
    assert res["config"]["global_batch"] == 8 * 32 and res["config"]["parallelism"] == "dp8" and res["scaling"] == "weak"
    assert sorted(d["rank"] for d in res["devices"]) == list(range(8)) and len({d["pid"] for d in res["devices"]}) == 8
    rk = res["rank_ms_per_step"]["per_rank"]
    assert len(rk) == 8 and rk[7] == max(rk) and rk[7] > rk[0]


def test_bench_failures_are_one_json_line():
    """No GPU / fewer GPUs than asked for: a one-line JSON error and a non-zero exit code, not a usage string."""
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-box behaviour")
    rc, res = _bench(["--gpus", "8"])
    assert rc != 0 and "error" in res and res["n_gpus"] == 8 and res["devices_visible"] == 0 and res["value"] is None
    rc, res = _bench([])
    assert rc != 0 and "HIP device" in res["error"]
