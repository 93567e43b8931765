"""Tensor-parallel all-reduce launcher: two gloo ranks on CPU (the row-parallel
matmul + launch sequence of parallel_utils/tensor_parallel/layers.py:416-426),
and graph capture / replay on one GPU rank over RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.getcwd())
    import torch, torch.distributed as dist
    import __graft_entry__
    __graft_entry__.load_package()
    from vllm_amd.worker import GraphAllReduce, initialize_all_reduce_launcher, get_all_reduce_launcher
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    hidden, shard, max_tokens = 64, 48, 40
    initialize_all_reduce_launcher(max_tokens, hidden, torch.float32)
    launcher = get_all_reduce_launcher()
    assert launcher.disable_graph and launcher.buffer.shape == (max_tokens, hidden)
    g = torch.Generator().manual_seed(7)
    x_full = torch.randn(24, shard * world, generator=g)        # same on every rank
    w_full = torch.randn(shard * world, hidden, generator=g)
    for num_tokens in (8, 24, 16):
        x = x_full[:num_tokens, rank * shard:(rank + 1) * shard]     # row-parallel: each rank holds a slice of K
        out = launcher.buffer[:num_tokens]
        torch.matmul(x, w_full[rank * shard:(rank + 1) * shard], out=out)
        y = launcher.launch(out)
        assert y.data_ptr() == launcher.buffer.data_ptr()
        assert torch.allclose(y, x_full[:num_tokens] @ w_full, atol=1e-4), (rank, num_tokens)
    try:
        launcher.launch(torch.zeros(8, hidden))
        raise SystemExit("a foreign tensor was accepted")
    except ValueError:
        pass
    dist.barrier()
    if rank == 0:
        print("ALL_REDUCE_OK")
""")


def test_two_rank_row_parallel_all_reduce_on_gloo():
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix="_tp_worker.py", delete=False) as f:
        f.write(_WORKER)
        path = f.name
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), path]
    try:
        res = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True,
                             text=True, timeout=240)
    finally:
        os.unlink(path)
    assert res.returncode == 0, res.stderr[-3000:]
    assert "ALL_REDUCE_OK" in res.stdout


def test_single_rank_launcher_is_a_no_op(pkg):
    import torch
    from vllm_amd.worker import GraphAllReduce
    launcher = GraphAllReduce(16, 8, torch.float16, device="cpu")
    assert launcher.buffer is None and launcher.world_size == 1
    x = torch.ones(4, 8)
    assert launcher.launch(x) is x


@pytest.mark.gpu
def test_graph_replay_over_rccl_single_rank_group(pkg):
    """A one-rank RCCL group has nothing to sum (RCCL elides the collective, the
    captured graph is empty): this covers communicator set-up and the launcher's
    capture-on-first-use bookkeeping on the GPU, and that the buffer contents
    survive graph construction.  The sum itself is covered by the gloo test."""
    import torch
    import torch.distributed as dist
    from vllm_amd.worker import GraphAllReduce
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", world_size=1, rank=0)
    try:
        launcher = GraphAllReduce(64, 128, torch.float16, device="cuda:0")
        assert launcher.world_size == 1 and launcher.buffer is None
        # force the multi-rank code path on the one-rank group
        launcher.world_size = 2
        launcher.buffer = torch.empty(64, 128, dtype=torch.float16, device="cuda:0")
        assert not launcher.disable_graph
        for n in (8, 32, 8):
            ref = torch.randn(n, 128, dtype=torch.float16, device="cuda:0")
            launcher.buffer[:n].copy_(ref)
            y = launcher.launch(launcher.buffer[:n])
            torch.cuda.synchronize()
            assert torch.equal(y, ref)
        assert sorted(launcher.graphs) == [8, 32]
    finally:
        dist.destroy_process_group()
