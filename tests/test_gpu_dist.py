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


def _torchrun(script_args, port, extra_env=None, timeout=600):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
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
