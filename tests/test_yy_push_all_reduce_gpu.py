"""GPU tests of the one-shot all-reduce (pa_comm_*, worker/push_all_reduce.py).

A one-GPU box is enough to check the algorithm: several ranks of a group live on the same
card, each with a stream of its own (peers reached by pointer), or in processes of their own
(peers reached through IPC mappings).  Expected value: the ranks' buffers added in rank order
in fp32, rounded once — bit-exact, and identical on every rank."""
import os
import textwrap

import pytest

from tests.test_all_reduce import _run_ranks

pytestmark = pytest.mark.gpu
_STREAMS = {}


def _expected(parts, dtype):
    import torch
    acc = torch.zeros(parts[0].shape, dtype=torch.float32, device=parts[0].device)
    for p in parts:
        acc = acc + p.float()
    return acc.to(dtype)


def _bits(t):
    import torch
    return t.view(torch.int16 if t.element_size() == 2 else torch.int32)


@pytest.mark.parametrize("algorithm", ["one_shot", "two_shot"])
@pytest.mark.parametrize("world", [2])
def test_ranks_on_one_gpu_by_pointer(pkg, world, algorithm):
    import torch
    from vllm_amd.worker import PushAllReduce
    dev = torch.device("cuda:0")
    max_bytes = 1 << 20
    ranks = PushAllReduce.local_ranks(world, max_bytes, algorithm={"one_shot": PushAllReduce.ONE_SHOT,
                                                                   "two_shot": PushAllReduce.TWO_SHOT}[algorithm])
    # the same streams for every case of this module: the first ones a process creates get hardware
    # queues of their own, later ones share
    streams = _STREAMS.setdefault(world, [torch.cuda.Stream(dev) for _ in range(world)])
    try:
        g = torch.Generator(device="cpu").manual_seed(world)
        k = 0
        for dtype in (torch.float16, torch.bfloat16, torch.float32):
            es = torch.tensor([], dtype=dtype).element_size()
            for n in (8, 4096, 40 * 4096, max_bytes // es, 24):   # odd and even reduces, growing and shrinking
                parts = [torch.randn(n, generator=g).to(dtype).to(dev) for _ in range(world)]
                want = _expected(parts, dtype)
                torch.cuda.synchronize()
                for r in range(world):
                    with torch.cuda.stream(streams[r]):
                        ranks[r](parts[r])
                torch.cuda.synchronize()
                k += 1
                if any(r.status()[1] for r in ranks):
                    # the ranks' kernels wait for one another: they must be on the GPU together, which
                    # streams of one process are only if the runtime gave them hardware queues of their own
                    pytest.skip(f"the {world} streams did not run concurrently (reduce {k}: waits given up "
                                f"{[r.status()[1] for r in ranks]}); the multi-process test covers {world} ranks")
                for r in range(world):
                    assert torch.equal(_bits(parts[r]), _bits(want)), (dtype, n, r)
        for r in ranks:
            assert r.status() == (k, 0)
        with pytest.raises(pkg._capi.PagedAttnError, match="exceeds"):
            ranks[0](torch.zeros(max_bytes // 2 + 8, dtype=torch.float32, device=dev))
        with pytest.raises(pkg._capi.PagedAttnError, match="multiple of 16"):
            ranks[0](torch.zeros(4, dtype=torch.float16, device=dev))
    finally:
        for r in ranks:
            r.close()


def test_a_peer_that_never_arrives_costs_one_bounded_wait(pkg):
    """Rank 1 of two never launches: rank 0's first reduce gives up after the wait budget (two seconds
    here; the serving default is 30), counts it and returns NaNs, every later reduce of that communicator
    does the same without waiting, and check() — what the launcher calls at the end of a step — raises."""
    import time
    import torch
    from vllm_amd.worker import GraphAllReduce, PushAllReduce
    dev = torch.device("cuda:0")
    assert PushAllReduce.DEFAULT_WAIT_SECONDS >= 30.0
    ranks = PushAllReduce.local_ranks(2, 1 << 16, wait_seconds=2.0)
    try:
        x = torch.ones(4096, dtype=torch.float16, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ranks[0](x)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        done, gave_up = ranks[0].status()
        assert done == 1 and gave_up >= 1, (done, gave_up)
        assert 1.5 < first < 6.0, first
        assert bool(torch.isnan(x).all())          # never a plausible-looking partial sum
        x.fill_(1.0)
        t0 = time.perf_counter()
        for _ in range(20):
            ranks[0](x)
        torch.cuda.synchronize()
        later = time.perf_counter() - t0
        done2, gave_up2 = ranks[0].status()
        assert done2 == 21 and gave_up2 > gave_up, (done2, gave_up2)
        assert later < 0.5, later
        assert bool(torch.isnan(x).all())
        with pytest.raises(RuntimeError, match="gave up"):
            ranks[0].check()
        launcher = GraphAllReduce(8, 512, torch.float16, device=dev, reduce_fn=ranks[0], disable_graph=True)
        with pytest.raises(RuntimeError, match="gave up"):
            launcher.check()
        ranks[1].check()  # the rank that never reduced has nothing to report
        with pytest.raises(pkg._capi.PagedAttnError, match="budget"):
            pkg._capi.comm_set_wait_budget(ranks[0]._comm, 0.0)
    finally:
        for r in ranks:
            r.close()


_IPC_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.getcwd())
    import torch, torch.distributed as dist
    import __graft_entry__
    __graft_entry__.load_package()
    from vllm_amd.worker import GraphAllReduce, PushAllReduce
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)                       # every rank on the one card
    dev = torch.device("cuda:0")
    tokens, hidden = 64, 512
    reduce = PushAllReduce(1024 * hidden * 2)
    g = torch.Generator(device="cpu").manual_seed(11)
    bare = 0
    # both forms, switched between reduces of one communicator; 1024 x 512 fp16 = 1 MiB is where
    # AUTO takes the two-shot form with four and eight ranks
    for algorithm in (PushAllReduce.ONE_SHOT, PushAllReduce.TWO_SHOT, PushAllReduce.AUTO, PushAllReduce.TWO_SHOT,
                      PushAllReduce.ONE_SHOT):
        reduce.algorithm = algorithm
        for n, dtype in ((64, torch.float16), (1024, torch.bfloat16), (8, torch.float16), (3, torch.float32),
                         (1000, torch.float16)):
            parts = [torch.randn(n, hidden, generator=g).to(dtype) for _ in range(world)]   # same on every rank
            want = sum((p.float() for p in parts[1:]), parts[0].float()).to(dtype)
            x = parts[rank].to(dev)
            reduce(x)
            torch.cuda.synchronize()
            bare += 1
            assert torch.equal(x.cpu().view(torch.int32), want.view(torch.int32)), (rank, algorithm, n, dtype)
    reduce.algorithm = PushAllReduce.AUTO
    # inside the launcher, as the tensor-parallel step uses it: side stream, events, one HIP graph per
    # (buffer, token count) captured on first use with live partial sums, replayed with fresh ones
    launcher = GraphAllReduce(tokens, hidden, torch.float16, device=dev, reduce_fn=reduce)
    assert not launcher.disable_graph
    for step, n in enumerate((16, 64, 16, 64, 16, 64, 40)):
        parts = [torch.randn(n, hidden, generator=g).half() for _ in range(world)]
        want = sum((p.float() for p in parts[1:]), parts[0].float()).half()
        buf = launcher.next_buffer(n)
        buf.copy_(parts[rank])
        pending = launcher.launch_async(buf)
        torch.cuda._sleep(200000)                      # the caller's stream has other work meanwhile
        y = pending.wait()
        torch.cuda.synchronize()
        assert torch.equal(y.cpu().view(torch.int16), want.view(torch.int16)), (rank, "launcher", step)
    # 16 and 40 tokens met buffer 0 of the ring, 64 buffer 1: three graphs, each captured after one warm-up reduce
    assert sorted(launcher.graphs) == [(0, 16), (0, 40), (1, 64)], sorted(launcher.graphs)
    done, timeouts = reduce.status()
    # set-up ran the self-test: both forms on a small and on a 1 MiB message, twice each
    assert reduce.self_test_reduces == 8, reduce.self_test_reduces
    assert (done, timeouts) == (reduce.self_test_reduces + bare + 7 + 3, 0), (rank, done, timeouts)
    launcher.check()
    dist.barrier()
    reduce.close()
    print(f"rank {rank} ok", flush=True)
""")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ranks_in_processes_of_their_own_through_ipc(pkg, world):
    res = _run_ranks(_IPC_WORKER, world, timeout=240, env_extra={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    for r in range(world):
        assert f"rank {r} ok" in res.stdout


def test_ipc_set_up_without_the_dmabuf_variable_works_or_names_it(pkg):
    """The ranks' processes map each other's memory through HIP IPC handles.  On hosts whose driver only
    offers dmabuf IPC that needs HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment; elsewhere it does not.
    Without the variable, set-up either works (and the reduces are right) or raises on EVERY rank with a
    message that names the variable — never a hang, never one rank failing alone."""
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    res = _run_ranks(_IPC_WORKER, 2, timeout=240, env_replace=env)
    if res.returncode == 0:
        assert "rank 0 ok" in res.stdout and "rank 1 ok" in res.stdout
    else:
        text = res.stdout + res.stderr
        assert text.count("HSA_ENABLE_IPC_MODE_LEGACY") >= 2, text[-4000:]
        assert "one-shot all-reduce set-up failed" in text, text[-4000:]


def test_bench_head_sharded_step_with_the_push_collective_two_ranks_on_one_gpu(pkg):
    """bench.py --tp as the driver launches it (torch.distributed.run, one line from rank 0), with both ranks
    on the one card: rendezvous over gloo, the all-reduce through the inboxes, replayed from HIP graphs."""
    import json
    import subprocess
    import sys
    from tests.test_all_reduce import ROOT, _free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--tp", "--steps", "2",
           "--warmup", "1", "--layers", "2", "--batch", "32", "--ctx", "256", "--dist-backend", "gloo",
           "--tp-collective", "push", "--tp-graph", "--no-c2", "--no-cpu-baseline", "--no-roofline"]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["tp"]["tp"] == 2 and "push" in d["tp"]["all_reduce_collective"]
    assert d["tp"]["push_waits_given_up"] == 0 and d["tp"]["push_reduces_done"] > 0


@pytest.mark.parametrize("deadline", [180.0, 0.001])
def test_bench_default_multi_rank_run_reports_both_head_sharded_legs_or_gives_up_on_time(pkg, deadline):
    """bench.py --gpus 2 without --tp (what the driver's scaling run launches), both ranks on the one card:
    the line carries the replica measurement plus the head-sharded step with the backend's all-reduce ("tp")
    and with the push all-reduce ("tp_push").  With a deadline that has passed before the extra legs start,
    the line still comes out, once, from rank 0, with the legs marked unfinished, and every rank exits 0."""
    import json
    import subprocess
    import sys
    from tests.test_all_reduce import ROOT, _free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--layers", "2", "--batch", "32", "--ctx", "256", "--dist-backend", "gloo",
           "--no-c2", "--no-cpu-baseline", "--no-roofline", "--tp-leg-deadline", str(deadline)]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    if deadline > 1:
        assert d["tp"]["tp"] == 2 and d["tp"]["value"] > 0, d["tp"]
        assert d["tp_push"]["push_waits_given_up"] == 0 and d["tp_push"]["push_reduces_done"] > 0, d["tp_push"]
    else:
        assert "deadline" in d["tp"]["error"] and "deadline" in d["tp_push"]["error"]


def test_bench_default_multi_rank_run_survives_a_rank_that_dies_in_the_push_leg(pkg):
    """The push leg of a default multi-rank run is measured by child processes: when rank 1's child aborts
    (what a GPU fault does to a process), the line still comes out once, from rank 0, with the replica
    measurement and the backend leg in it and the push leg marked failed, and every parent rank exits 0."""
    import json
    import subprocess
    import sys
    from tests.test_all_reduce import ROOT, _free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--layers", "2", "--batch", "32", "--ctx", "256", "--dist-backend", "gloo",
           "--no-c2", "--no-cpu-baseline", "--no-roofline", "--tp-leg-deadline", "90", "--rehearse-child-abort", "1"]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["tp"]["tp"] == 2 and d["tp"]["value"] > 0, d["tp"]
    assert "error" in d["tp_push"] and "child" in d["tp_push"]["error"], d["tp_push"]
