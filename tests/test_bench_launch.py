"""bench.py's multi-rank launch path: `python bench.py --gpus 2` must start its own ranks when no
launcher is around it, and must also run under torch.distributed.run exactly as the driver starts it.  `--dry` keeps the
GPU out of it (gloo on the host), so this runs in the CPU suite."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_bare_invocation_spawns_its_own_ranks():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry",
                          "--scaling", "weak"], capture_output=True, text=True, timeout=300, env=_env())
    assert res.returncode == 0, res.stderr[-2000:]
    out = _one_json_line(res.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["dry"] is True
    # opt-in: every rank keeps 8 images (BASELINE config 3 = 4 x 8) -> weak scaling, whole-job value
    assert out["config"]["per_gpu_batch"] == 8 and out["config"]["global_batch"] == 16 and out["scaling"] == "weak"


def test_under_the_drivers_launcher():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_env())
    assert res.returncode == 0, res.stderr[-2000:]
    out = _one_json_line(res.stdout)
    # default since round 3: north_star's partition -- the metric's GLOBAL batch of 8 split over the ranks
    # (DGDE/data/build.py:63-67) -> strong scaling
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["per_gpu_batch"] == 4 and out["config"]["global_batch"] == 8


def test_single_rank_dry():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--steps", "2", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=_env())
    assert res.returncode == 0, res.stderr[-2000:]
    assert _one_json_line(res.stdout)["n_gpus"] == 1
