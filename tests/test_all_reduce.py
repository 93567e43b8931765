"""Tensor-parallel all-reduce launcher and the head-sharded decode step.

CPU, two gloo ranks: the row-parallel matmul + launch sequence of
parallel_utils/tensor_parallel/layers.py:416-426, and the sharded decode step
(each rank's heads through stand-in operators backed by the oracle): the ranks'
heads side by side equal the unsharded result, the summed o_proj equals the
full one, and the hand-off protocol ran in order.
GPU: the side-stream / event / HIP-graph machinery on one GPU with a local
stand-in for the collective (a one-rank RCCL group would capture an empty
graph), and the real collective on two ranks where two GPUs exist."""
import os
import socket
import subprocess
import sys
import tempfile
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(source: str, nproc: int, timeout: int = 300, env_extra=None, env_replace=None):
    with tempfile.NamedTemporaryFile("w", suffix="_tp_worker.py", delete=False) as f:
        f.write(source)
        path = f.name
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), path]
    env = dict(os.environ if env_replace is None else env_replace, OMP_NUM_THREADS="1")
    env.update(env_extra or {})
    try:
        return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    finally:
        os.unlink(path)


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
    initialize_all_reduce_launcher(max_tokens, hidden, torch.float32)  # device follows the group's backend: gloo -> CPU
    launcher = get_all_reduce_launcher()
    assert launcher.device.type == "cpu" and launcher.disable_graph and launcher.buffer.shape == (max_tokens, hidden)
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
    # the ring: two reduces in flight, the third hand-out of a buffer fences its first reduce
    a = launcher.next_buffer(8); a.fill_(rank + 1.0); pa = launcher.launch_async(a)
    b = launcher.next_buffer(8); b.fill_(10.0 * (rank + 1)); pb = launcher.launch_async(b)
    assert a.data_ptr() != b.data_ptr() and not pa.waited and not pb.waited
    total = sum(range(1, world + 1))
    c = launcher.next_buffer(8)
    assert c.data_ptr() == a.data_ptr() and pa.waited and not pb.waited
    assert torch.all(pb.wait() == 10.0 * total)
    # overwriting partial sums that were never waited for is refused
    d = launcher.buffers[0][:8]; d.fill_(1.0); pd = launcher.launch_async(d)
    try:
        launcher.launch_async(d)
        raise SystemExit("an un-waited buffer was reduced again")
    except RuntimeError:
        pd.wait()
    dist.barrier()
    if rank == 0:
        print("ALL_REDUCE_OK")
""")


def test_two_rank_row_parallel_all_reduce_on_gloo():
    res = _run_ranks(_WORKER, 2)
    assert res.returncode == 0, res.stderr[-3000:]
    assert "ALL_REDUCE_OK" in res.stdout


_SHARDED_STEP = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.getcwd())
    import numpy as np
    import torch, torch.distributed as dist
    import __graft_entry__
    __graft_entry__.load_package()
    from oracle import datagen, pa_oracle
    from vllm_amd.worker import GraphAllReduce, TensorParallelDecodeStep, head_shard

    class OracleOps:
        # CPU stand-ins with the operator modules' signatures (test infrastructure)
        @staticmethod
        def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping):
            pa_oracle.reshape_and_cache(key.numpy(), value.numpy(), key_cache.numpy(), value_cache.numpy(),
                                        slot_mapping.numpy(), "float32")
        @staticmethod
        def single_query_cached_kv_attention(out, query, key_cache, value_cache, scale, block_tables, context_lens,
                                             block_size, max_context_len):
            out.copy_(torch.from_numpy(pa_oracle.single_query_cached_kv_attention(
                query.numpy(), key_cache.numpy(), value_cache.numpy(), scale, block_tables.numpy(),
                context_lens.numpy(), "float32")))

    dist.init_process_group(backend="gloo")
    rank, tp = dist.get_rank(), dist.get_world_size()
    S, H, D, B, L, hidden = 9, 8, 64, 16, 3, 96
    case = datagen.DecodeCase(name="tp", dtype="float32", num_seqs=S, num_heads=H, head_size=D, block_size=B,
                              num_blocks=40, max_context=70, dist="normal", tables="forked", fork=3, seed=88)
    lo, hi = head_shard(rank, tp, H)
    launcher = GraphAllReduce(16, hidden, torch.float32, device="cpu", record_trace=True)
    step = TensorParallelDecodeStep(rank, tp, H, D, D ** -0.5, launcher, attention_ops=OracleOps, cache_ops=OracleOps)
    g = torch.Generator().manual_seed(3)
    w_o = torch.randn(L, H * D, hidden, generator=g) * (H * D) ** -0.5
    pendings, want_outs, want_caches = [], [], []
    locals_ = []
    for l in range(L):
        case.seed = 88 + l
        inp = datagen.make_decode_inputs(case)          # same on every rank
        ctx = inp["context_lens"].astype(np.int64)
        slots = (inp["block_tables"][np.arange(S), (ctx - 1) // B].astype(np.int64) * B + (ctx - 1) % B).astype(np.int32)
        qkv = torch.from_numpy(inp["qkv"])
        # unsharded result on every rank, for comparison
        kc_full, vc_full = inp["key_cache"].copy(), inp["value_cache"].copy()
        pa_oracle.reshape_and_cache(inp["qkv"][:, 1], inp["qkv"][:, 2], kc_full, vc_full, slots, "float32")
        full = torch.from_numpy(pa_oracle.single_query_cached_kv_attention(
            inp["qkv"][:, 0], kc_full, vc_full, inp["scale"], inp["block_tables"], inp["context_lens"], "float32"))
        want_outs.append(full)
        # this rank's shard: its heads of qkv (column slices of the LOCAL fused projection) and of the caches
        qkv_local = qkv[:, :, lo:hi].contiguous()
        kc = torch.from_numpy(np.ascontiguousarray(inp["key_cache"][:, lo:hi]))
        vc = torch.from_numpy(np.ascontiguousarray(inp["value_cache"][:, lo:hi]))
        out_local = torch.empty(S, hi - lo, D)
        if l == 0:
            step.begin_step()
        p = step.layer(qkv_local[:, 0], qkv_local[:, 1], qkv_local[:, 2], kc, vc, torch.from_numpy(slots),
                       torch.from_numpy(inp["block_tables"]), torch.from_numpy(inp["context_lens"]),
                       inp["max_context_len"], w_o[l, lo * D:hi * D], out_local, int(ctx.sum()))
        pendings.append(p)
        locals_.append(out_local)
        assert np.array_equal(kc.numpy(), kc_full[:, lo:hi]) and np.array_equal(vc.numpy(), vc_full[:, lo:hi])
        assert torch.equal(out_local, full[:, lo:hi]), (rank, l)
    # the heads of all ranks side by side are the unsharded output
    for l in range(L):
        parts = [torch.empty_like(locals_[l]) for _ in range(tp)]
        dist.all_gather(parts, locals_[l])
        assert torch.equal(torch.cat(parts, dim=1), want_outs[l]), l
    # layer 0's reduce was fenced when layer 2 took its buffer back; the others are still pending
    assert [p.waited for p in pendings] == [True, False, False][:L]
    # sums: buffers are reused, so check the last two layers (still intact) against the full o_proj
    for l in (L - 2, L - 1):
        want = want_outs[l].reshape(S, H * D) @ w_o[l]
        got = pendings[l].wait()
        assert torch.allclose(got, want, atol=1e-4, rtol=1e-4), (rank, l, float((got - want).abs().max()))
    # protocol order: per buffer, ready(n) < reduced(n) < waited(n) <= ready(n + 2)
    t = launcher.trace
    for seq in range(1, L + 1):
        i_ready = t.index(("ready", (seq - 1) % 2, seq))
        i_red = t.index(("reduced", (seq - 1) % 2, seq))
        i_wait = t.index(("waited", (seq - 1) % 2, seq))
        assert i_ready < i_red < i_wait
        if seq + 2 <= L:
            assert i_wait < t.index(("ready", (seq - 1) % 2, seq + 2))
    dist.barrier()
    if rank == 0:
        print("SHARDED_STEP_OK")
""")


def test_two_rank_head_sharded_decode_step_on_gloo():
    res = _run_ranks(_SHARDED_STEP, 2)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert "SHARDED_STEP_OK" in res.stdout


def test_head_shard_bounds(pkg):
    from vllm_amd.worker import head_shard
    assert [head_shard(r, 4, 64) for r in range(4)] == [(0, 16), (16, 32), (32, 48), (48, 64)]
    assert head_shard(1, 2, 40) == (20, 40) and head_shard(0, 1, 12) == (0, 12)
    with pytest.raises(ValueError):
        head_shard(0, 3, 40)
    with pytest.raises(ValueError):
        head_shard(2, 2, 40)


def test_single_rank_launcher_is_a_no_op(pkg):
    import torch
    from vllm_amd.worker import GraphAllReduce
    launcher = GraphAllReduce(16, 8, torch.float16, device="cpu")
    assert launcher.buffer is None and launcher.world_size == 1
    x = torch.ones(4, 8)
    assert launcher.launch(x) is x
    assert launcher.launch_async(x).wait() is x


@pytest.mark.gpu
def test_side_stream_graph_replay_with_a_stand_in_collective(pkg):
    """The launcher's GPU machinery — side stream, ready/done events, graphs
    captured on the side stream per (buffer, token count), on first use with
    live partial sums in the buffer and ahead of time, the two-buffer ring —
    driven on one GPU by a stand-in collective that adds a fixed "remote"
    partial sum (a one-rank RCCL group would reduce nothing and capture an empty
    graph).  The caller's stream is kept busy with a long kernel queue between
    launch_async and wait, so a missing fence shows up as a wrong sum."""
    import torch
    from vllm_amd.worker import GraphAllReduce
    dev = torch.device("cuda:0")
    hidden, max_tokens = 512, 64
    remote = torch.randn(max_tokens, hidden, device=dev).to(torch.float16)

    def stand_in(t):
        t.add_(remote[:t.shape[0]])

    launcher = GraphAllReduce(max_tokens, hidden, torch.float16, device=dev, reduce_fn=stand_in, record_trace=True)
    assert not launcher.disable_graph and launcher.buffer is launcher.buffers[0]
    launcher.precapture([8, 16])
    assert sorted(launcher.graphs) == [(0, 8), (0, 16), (1, 8), (1, 16)]
    busy = torch.randn(2048, 2048, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for n in (8, 32, 16, 32, 8, 64):  # 32 and 64 are captured on first use, with live data in the buffer
        local = torch.randn(n, hidden, device=dev, generator=g).to(torch.float16)
        for _ in range(3):
            busy = busy @ busy * 1e-3  # the producer of the partial sums sits behind a queue of work
        buf = launcher.next_buffer(n)
        buf.copy_(local)
        pending = launcher.launch_async(buf)
        for _ in range(3):
            busy = busy @ busy * 1e-3  # work that overlaps the reduce
        y = pending.wait()
        want = local + remote[:n]
        assert torch.equal(y, want), n  # compared on the caller's stream, ordered behind the wait
    assert {k[1] for k in launcher.graphs} == {8, 16, 32, 64}
    # the reference's contract on `buffer`: launch() is ordered on the current stream
    for n in (8, 24):
        local = torch.randn(n, hidden, device=dev, generator=g).to(torch.float16)
        launcher.buffer[:n].copy_(local)
        y = launcher.launch(launcher.buffer[:n])
        assert y.data_ptr() == launcher.buffer.data_ptr() and torch.equal(y, local + remote[:n])
    torch.cuda.synchronize()
    kinds = [k for k, _, _ in launcher.trace]
    assert kinds.count("ready") == kinds.count("reduced") == kinds.count("waited") == 8


_RCCL_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.getcwd())
    import torch, torch.distributed as dist
    import __graft_entry__
    __graft_entry__.load_package()
    from vllm_amd.worker import GraphAllReduce, PushAllReduce
    rank = int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(rank)
    dist.init_process_group(backend="nccl")
    world = dist.get_world_size()
    dev = torch.device("cuda", rank)
    hidden = 1024

    def partial_sums(n, seed):
        # every rank can compute every rank's partial sums: the expected total needs no second collective
        g = torch.Generator(device="cpu").manual_seed(seed)
        return [torch.randn(n, hidden, generator=g).half() for _ in range(world)]

    def drive(launcher, exact, label):
        busy = torch.randn(1024, 1024, device=dev)
        # 8 is pre-captured; 32 and 48 are captured on first use, in the middle of a step, with LIVE partial
        # sums in the buffer (the warm-up reduce of the capture must not eat them); every count comes round
        # on both buffers of the ring, and each graph is replayed with fresh data
        for step, n in enumerate((8, 32, 8, 32, 48, 48, 8, 32, 64)):
            parts = partial_sums(n, 100 + step)
            total = parts[0].float()
            for p in parts[1:]:
                total = total + p.float()
            for _ in range(2):
                busy = busy @ busy * 1e-3            # the producer of the partial sums sits behind queued work
            buf = launcher.next_buffer(n)
            buf.copy_(parts[rank].to(dev), non_blocking=True)
            pending = launcher.launch_async(buf)
            for _ in range(2):
                busy = busy @ busy * 1e-3            # work that overlaps the reduce
            y = pending.wait()
            torch.cuda.synchronize()
            if exact:   # rank-ordered fp32 sum rounded once: the push all-reduce's contract
                assert torch.equal(y.cpu(), total.half()), (label, rank, step, n)
            else:       # RCCL chooses its own order and intermediate precision
                assert torch.allclose(y.cpu().float(), total, atol=2e-2, rtol=2e-3), (label, rank, step, n)
        launcher.check()

    # 1. the backend's collective (RCCL) captured and replayed on the side stream
    launcher = GraphAllReduce(64, hidden, torch.float16, device=dev)
    assert not launcher.disable_graph
    launcher.precapture([8])
    assert sorted(launcher.graphs) == [(0, 8), (1, 8)]
    drive(launcher, False, "rccl graph")
    assert {k[1] for k in launcher.graphs} == {8, 32, 48, 64}
    # 2. the same without graphs: direct calls on the side stream (bench.py's default for its RCCL leg)
    drive(GraphAllReduce(64, hidden, torch.float16, device=dev, disable_graph=True), False, "rccl direct")
    # 3. the one-shot push all-reduce between two GPUs: peers' memory through IPC over xGMI.  Set-up runs the
    # collective self-test (both forms, small and multi-MiB messages); then the launcher, graphs on live data.
    push = PushAllReduce(4 << 20, device=dev, wait_seconds=20.0)
    assert push.self_test_reduces == 8
    launcher = GraphAllReduce(64, hidden, torch.float16, device=dev, reduce_fn=push)
    launcher.precapture([8])
    drive(launcher, True, "push graph")
    done, gave_up = push.status()
    assert gave_up == 0 and done >= 8 + 9, (done, gave_up)
    torch.cuda.synchronize()
    dist.barrier()
    push.close()
    dist.destroy_process_group()
    if rank == 0:
        print("RCCL_GRAPH_OK")
""")


@pytest.mark.gpu
def test_graph_replay_over_rccl_two_ranks():
    """The real collective, captured and replayed on the side stream — pre-captured and captured on first
    use with live partial sums in the buffer — and issued directly; then this package's push all-reduce
    between the two GPUs, self-test included.  Needs two GPUs (one process each)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs; the single-GPU test above covers the launcher's stream and graph machinery")
    res = _run_ranks(_RCCL_WORKER, 2, env_extra={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert "RCCL_GRAPH_OK" in res.stdout


_PUSH_SETUP_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.getcwd())
    import torch, torch.distributed as dist
    import __graft_entry__
    __graft_entry__.load_package()
    from vllm_amd.worker import PushAllReduce
    dist.init_process_group(backend="gloo")
    rank = dist.get_rank()
    # no GPU here: creating the communicator fails, and it must fail on every rank together
    try:
        PushAllReduce(1 << 16, device=torch.device("cuda", 0))
        raise SystemExit("set-up succeeded without a GPU")
    except RuntimeError as e:
        assert "set-up failed on rank 0" in str(e), str(e)
    dist.barrier()    # the group is still in step
    print(f"rank {rank} ok", flush=True)
""")


def test_push_all_reduce_set_up_fails_on_all_ranks_together():
    """Without a GPU the communicator cannot be created; every rank learns so in the same collective and
    raises, instead of one rank raising and the others waiting in the next collective."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: set-up would succeed")
    res = _run_ranks(_PUSH_SETUP_WORKER, 2, timeout=120)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert "rank 0 ok" in res.stdout and "rank 1 ok" in res.stdout
