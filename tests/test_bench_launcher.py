"""bench.py started plainly with --gpus N > 1 must launch its own workers: the parent spawns
`python -m torch.distributed.run --nproc-per-node N` on 127.0.0.1 before any GPU call and relays rank 0's JSON line.
Rehearsed here on the CPU: `--dry-launch` runs the same launcher and the workers' rendezvous / barrier / max-over-ranks
over gloo, without kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)


def test_plain_invocation_spawns_workers_and_relays_one_line():
    p = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["world_size_seen"] == 2 and res["steps"] == 3 and res["warmup"] == 1
    assert res["launcher"].startswith("bench.py spawned")
    assert res["ms_per_step"] >= 20.0          # max over ranks: rank 1 sleeps 20 ms, rank 0 10 ms
    assert "torch.distributed.run" in p.stderr and "--master-addr 127.0.0.1" in p.stderr


def test_under_a_launcher_the_worker_does_not_spawn_again():
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2 and "launcher" not in json.loads(lines[0])


def test_world_size_mismatch_is_an_error():
    p = _run("--gpus", "1", "--dry-launch", env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)
