"""CPU, world_size 2, gloo: the multi-process plumbing of bench.py (rank env, barrier-bracketed timing, MAX over ranks,
whole-job value, one JSON line from rank 0).  The data path itself needs no collective in forward."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_gloo():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5",
           "--warmup", "1", "--dry-run-cpu", "--batch", "8"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 5 and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 16
    # rank 1 sleeps 4 ms per step, rank 0 2 ms: MAX over ranks -> >= 4 ms/step; whole-job value = 16 images / step time
    assert r["ms_per_step"] >= 3.9
    assert abs(r["value"] - 16 * 1e3 / r["ms_per_step"]) / r["value"] < 0.01
    # every rank's own bracket time travels in the line (a slow rank is visible): rank 1 sleeps twice as long per step
    assert len(r["rank_ms_per_step"]) == 2 and r["rank_ms_per_step"][1] >= 3.9 and max(r["rank_ms_per_step"]) <= r["ms_per_step"] + 1e-3


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the process starts two ranks of itself (gloo here)
    and the line says n_gpus 2 with the process group's own world size in it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                          "--dry-run-cpu"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["world_size"] == 2 and r["config"]["global_batch"] == 16
    assert "world size 2" in r["config"]["parallelism"] and r["ms_per_step"] >= 3.9


def test_bench_rejects_a_world_size_that_is_not_gpus():
    """--gpus is the contract: a launcher that started a different number of ranks is an error, not a silent n_gpus."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                          "--dry-run-cpu"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert out.returncode == 2 and "WORLD_SIZE=1" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_single_process_dry_run():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--dry-run-cpu"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["metric"].startswith("images/sec/GPU")
    assert "no collective" in r["config"]["parallelism"]


def test_gradient_average_two_ranks():
    """The training step's one collective (TrainEngine.reduce_gradients -> dist_utils.average_flat_) on 2 gloo ranks:
    bucketed contiguous all-reduce of a flat buffer, ragged last bucket, empty buffer."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "RANK 0 OK=True" in out.stdout and "RANK 1 OK=True" in out.stdout
    # dist_utils.GradExchange / bucket_ranges: the exchange overlapped with backward (same worker, second half)
    assert "RANK 0 EXCHANGE=True" in out.stdout and "RANK 1 EXCHANGE=True" in out.stdout
    # dist_utils.broadcast_tuner_choices: both ranks end with rank 0's kernel choices (same worker, third part)
    assert "RANK 0 TUNER=True" in out.stdout and "RANK 1 TUNER=True" in out.stdout
