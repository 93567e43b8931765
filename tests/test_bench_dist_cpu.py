"""World-size-2 test of bench.py's multi-process plumbing on CPU (gloo):
rendezvous from the torch.distributed.run environment, barrier fences,
MAX-over-ranks timing and the whole-job aggregate.  No kernels run."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_plumbing_aggregates_over_ranks():
    steps, batch = 5, 256
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "1",
           "--cpu-plumbing-test", "--batch", str(batch)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["scaling"] == "weak"
    # rank 1 sleeps 4 ms per step, rank 0 2 ms: the slower rank sets the time
    assert d["ms_per_step"] >= 3.9, d
    # value = all ranks' tokens / max time
    assert abs(d["value"] - 2 * batch * steps / (d["ms_per_step"] * steps / 1e3)) < 1e-6 * d["value"]


def test_single_process_plumbing():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "0",
                          "--cpu-plumbing-test"], cwd=ROOT, capture_output=True, text=True, timeout=120,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["value"] > 0


def test_algorithmic_bytes_match_the_baseline_table():
    sys.path.insert(0, ROOT)
    import bench
    # BASELINE.md section 3
    assert bench.algorithmic_bytes([1024] * 256, 12, 64, 16, 2) == 806_159_360
    assert bench.algorithmic_bytes([1024] * 256, 32, 128, 16, 2) == 4_299_228_160
