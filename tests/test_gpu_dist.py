"""Multi-GPU readiness checked on ONE GPU: the RCCL code paths -- process-group set-up,
barrier, MAX-reduce of the elapsed time, parameter broadcast, gradient-bucket all-reduce, the scalar num_objects
all-reduce -- run at world size 1 under torch.distributed.run with backend nccl (= RCCL on ROCm)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script_args, port, extra_env=None, timeout=600, nproc=1):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, "torchrun failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


def test_bench_under_rccl_world1():
    """bench.py launched exactly as the driver launches it for N > 1 (SQDET_FORCE_DIST=1 makes the one-rank job take the
    distributed path): init_process_group(nccl), barrier, all_reduce(MAX) of the time -- and the JSON contract."""
    out = _torchrun(["bench.py", "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], 29611, {"SQDET_FORCE_DIST": "1"})
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["steps"] == 5 and res["value"] > 1000 and res["scaling"] == "weak"
    assert res["roofline"]["frac"] > 0 and res["config"]["name"] == "sqdet_infer"


def test_training_bench_under_rccl_world1():
    """The N > 1 entry point of the training configs (gradient all-reduce over RCCL inside the step), one rank."""
    out = _torchrun(["bench.py", "--gpus", "1", "--config", "sqdet_train_fp32", "--batch", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                    29612, {"SQDET_FORCE_DIST": "1"})
    res = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["value"] > 10 and res["config"]["name"] == "sqdet_train_fp32"
    assert all(v == v and abs(v) < 1e6 for v in res["losses"].values())


def test_trainer_steps_under_rccl_equal_single_process():
    """Two trainer steps with the RCCL process group (broadcast, bucket all-reduce, scalar num_objects all-reduce at world
    size 1) leave the variables BITWISE equal to the same steps without a process group, in both loss normalisations."""
    out = _torchrun(["tests/dist_trainer_check.py"], 29613)
    line = [l for l in out.splitlines() if l.startswith("DIST_CHECK ")][-1]
    res = json.loads(line[len("DIST_CHECK "):])
    assert res["world"] == 1 and res["replica_mean_equal"] and res["global_equal"], res
    assert res["moved"] > 0 and res["loss_single"] == res["loss_dist"] == res["loss_global"], res


def test_world2_trainer_on_one_gpu():
    """N = 2 on the ONE GPU this box has: two ranks, both on cuda:0, backend gloo (RCCL refuses two ranks per device), the real
    SqueezeDetTrainer + GraphedStep -- constructor broadcast from rank 0, the eager num_objects all-reduce ahead of the graph
    replay, the SUM all-reduce of the flat device gradient bucket, both loss normalisations, dropout on (one mask per global
    batch, a replica takes its rows).  (i) both ranks end BITWISE identical in both modes; (ii) global-num_objects mode reproduces
    the single-process steps at batch 2B up to float32 summation order (the reference is single-device: train.py:107,302-304)."""
    out = _torchrun(["tests/dist2_trainer_check.py"], 29614, nproc=2)
    res = json.loads([l for l in out.splitlines() if l.startswith("DIST2_CHECK ")][-1][len("DIST2_CHECK "):])
    assert res["world"] == 2 and res["backend"] == "gloo", res
    for mode in ("global", "replica_mean"):
        assert res[mode + "_broadcast_ok"] and res[mode + "_ranks_bitwise_equal"], res
        assert res[mode + "_moved"] > 1e-4 and res[mode + "_skipped"] == 0, res
    assert res["global_max_abs_diff_vs_single_2B"] < 1e-6, res
    # (replica-mean normalises every replica by its OWN num_objects: a different graph from the global one, not compared)


def test_bench_two_ranks_share_one_gpu():
    """bench.py --gpus 2 with SQDET_SHARE_DEVICE=1: the self-launcher starts two ranks that share cuda:0 (gloo rendezvous,
    barrier, MAX-reduce of the time, rank census) -- the N > 1 bench path with real kernels on a one-GPU box.  The line says so
    (`shared_device`): its value is not a two-GPU figure."""
    env = dict(os.environ)
    env.update({"SQDET_SHARE_DEVICE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ranks_seen"] == 2 and res["n_gpus"] == 2 and res["shared_device"] is True, res
    assert len(res["devices"]) == 2 and res["devices"][0]["uuid"] == res["devices"][1]["uuid"], res["devices"]
    assert res["value"] > 1000 and res["config"]["global_batch"] == 64, res
