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
    # per launch: K and V of every token once, plus q, out, the block tables and context lengths
    assert bench.algorithmic_bytes([1024] * 256, 12, 64, 16, 2) == 806_159_360
    assert bench.algorithmic_bytes([1024] * 256, 32, 128, 16, 2) == 4_299_228_160


def test_two_rank_plumbing_tensor_parallel_counts_the_batch_once():
    """--tp: every rank works on the same tokens (strong scaling): value = batch * steps / max time."""
    steps, batch = 4, 64
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "1",
           "--cpu-plumbing-test", "--batch", str(batch), "--tp"]
    res = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True,
                         timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "TP=2" in d["metric"]
    assert abs(d["value"] - batch * steps / (d["ms_per_step"] * steps / 1e3)) < 1e-6 * d["value"]


def test_torch_cpu_baseline_statements_match_the_c_oracle():
    """The two torch-CPU statements that bench.py times as cpu_baseline legs
    compute what the C oracle computes (which the reference's own function pins)."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from oracle import datagen, pa_oracle, torch_cpu_baseline
    for dtype, B, D in (("float16", 16, 64), ("float32", 8, 80), ("bfloat16", 32, 128)):
        case = datagen.DecodeCase(name="cpu", dtype=dtype, num_seqs=6, num_heads=3, head_size=D, block_size=B,
                                  num_blocks=40, context_lens=[1, B, B + 1, 70, 0, 37], dist="normal",
                                  tables="forked", fork=2, seed=17)
        inp = datagen.make_decode_inputs(case)
        want = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
            inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
            inp["context_lens"], dtype), dtype)
        args = (datagen.storage_to_torch(inp["qkv"], dtype)[:, 0], datagen.storage_to_torch(inp["key_cache"], dtype),
                datagen.storage_to_torch(inp["value_cache"], dtype), torch.from_numpy(inp["block_tables"]),
                torch.from_numpy(inp["context_lens"]), inp["scale"])
        tol = {"float16": 2e-3, "float32": 1e-5, "bfloat16": 1.6e-2}[dtype]  # the oracle's result is rounded to dtype
        for fn in (torch_cpu_baseline.decode_loop, torch_cpu_baseline.decode_vectorised):
            got = fn(*args).numpy()
            assert np.allclose(got, want, atol=tol, rtol=tol), (dtype, fn.__name__, float(np.abs(got - want).max()))
