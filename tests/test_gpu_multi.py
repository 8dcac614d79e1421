"""N >= 2 ranks over RCCL, one rank per GPU (tools/train_net.py:83-88, 224-226: the reference wraps the model in
DistributedDataParallel over NCCL).  These tests turn themselves on when the box shows at least two GPUs and are skipped on the
one-GPU test boxes, where the same code paths run over gloo with both ranks on one device (tests/test_gpu_train.py) and with one
rank over RCCL (`oneRankReduce`).  The 8-GPU curve itself is the driver's (`SCALE_rNN.json`)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWO_GPUS = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least two GPUs (RCCL refuses two ranks on one device)")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(n, script_args, env=None, timeout=1500):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    e = dict(os.environ, OMP_NUM_THREADS="4", NCCL_DEBUG="VERSION", HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)


@TWO_GPUS
def test_two_ranks_over_rccl_average_gradients():
    """The real training engine on two GPUs: the bucketed all-reduce (RCCL AVG on the update stream behind events) must leave the
    average of the two ranks' locally computed gradients in the flat buffer, and train_step must apply the identical update on
    both ranks."""
    out = _launch(2, [os.path.join(ROOT, "tests", "dist_gpu_worker.py")], env=dict(OSD_DIST_BACKEND="nccl"), timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    for r in (0, 1):
        assert "RANK %d BACKEND=nccl WORLD=2" % r in out.stdout, out.stdout[-1500:]
        assert "RANK %d GPU_EXCHANGE=True" % r in out.stdout, out.stdout[-1500:]


@TWO_GPUS
def test_bench_two_ranks_over_rccl():
    """`bench.py --gpus 2` exactly as the driver launches it, over nccl (= RCCL): ONE line from rank 0, world size 2 seen by the
    process group, the backend and RCCL's version in the parallelism string, both ranks' own step times in the line, whole-job
    value = 16 images per step / MAX-over-ranks time."""
    out = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-conv-timing"])
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["world_size"] == 2 and r["config"]["global_batch"] == 16 and r["scaling"] == "weak"
    par = r["config"]["parallelism"]
    assert "backend nccl" in par and "RCCL" in par and "buckets" in par
    assert len(r["rank_ms_per_step"]) == 2 and all(t > 0 for t in r["rank_ms_per_step"])
    assert abs(r["value"] - 16 * 1e3 / r["ms_per_step"]) / r["value"] < 0.01
    assert "cpu_baseline" not in r


@TWO_GPUS
def test_bench_self_launch_over_rccl_uses_every_visible_gpu():
    """`python bench.py --gpus N` with no launcher, N = every GPU the box shows (2, 4 or 8): bench.py starts its own ranks."""
    n = min(8, torch.cuda.device_count())
    n = 8 if n >= 8 else (4 if n >= 4 else 2)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
                          "--no-conv-timing"], capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == n and r["config"]["world_size"] == n and len(r["rank_ms_per_step"]) == n


def test_multi_gpu_tests_are_armed():
    """On a one-GPU box the tests above are skipped, not absent: this one records what the box showed."""
    n = torch.cuda.device_count()
    assert n >= 1
    print("GPUs visible: %d -> RCCL N >= 2 tests %s" % (n, "RUN" if n >= 2 else "skipped"))
