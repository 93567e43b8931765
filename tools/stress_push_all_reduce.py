#!/usr/bin/env python3
"""Stress run of the one-shot / two-shot all-reduce: every rank a process of its own on cuda:0 (or
on its own GPU with --one-gpu-per-rank), thousands of reduces of random sizes, data types and
algorithms, every result checked bit for bit against the rank-ordered fp32 sum.

    python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 tools/stress_push_all_reduce.py [--reduces 2000]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import __graft_entry__  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reduces", type=int, default=2000)
    ap.add_argument("--max-bytes", type=int, default=2 << 20)
    ap.add_argument("--one-gpu-per-rank", action="store_true")
    ap.add_argument("--check-every", type=int, default=1, help="synchronise and compare every n-th reduce (all are compared at the end of a burst)")
    args = ap.parse_args()
    __graft_entry__.load_package()
    from vllm_amd.worker import PushAllReduce
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if args.one_gpu_per_rank else 0)
    torch.cuda.set_device(dev)
    reduce = PushAllReduce(args.max_bytes, device=dev)
    reduce.self_test()
    g = torch.Generator(device="cpu").manual_seed(1234)   # same stream of random numbers on every rank
    dtypes = (torch.float16, torch.bfloat16, torch.float32)
    algos = (PushAllReduce.AUTO, PushAllReduce.ONE_SHOT, PushAllReduce.TWO_SHOT)
    bad = 0
    pending = []
    for it in range(args.reduces):
        dtype = dtypes[int(torch.randint(0, 3, (1,), generator=g))]
        es = torch.tensor([], dtype=dtype).element_size()
        # sizes from 16 bytes up, log-uniform, multiples of 16 bytes
        nbytes = int(16 * max(1, int(2 ** (float(torch.rand(1, generator=g)) * 17.0))))
        nbytes = min(nbytes // 16 * 16, args.max_bytes)
        n = nbytes // es
        reduce.algorithm = algos[int(torch.randint(0, 3, (1,), generator=g))]
        parts = [torch.randn(n, generator=g).to(dtype) for _ in range(world)]
        want = parts[0].float()
        for p in parts[1:]:
            want = want + p.float()
        want = want.to(dtype)
        x = parts[rank].to(dev)
        reduce(x)
        pending.append((it, x, want, dtype, nbytes, reduce.algorithm))
        if (it + 1) % args.check_every == 0 or it + 1 == args.reduces:
            torch.cuda.synchronize()
            for (i, xi, wi, dt, nb, al) in pending:
                view = torch.int16 if xi.element_size() == 2 else torch.int32
                if not torch.equal(xi.cpu().view(view), wi.view(view)):
                    bad += 1
                    if bad <= 5:
                        print(f"rank {rank}: reduce {i} ({dt}, {nb} bytes, algorithm {al}) differs", flush=True)
            pending = []
        if rank == 0 and (it + 1) % 100 == 0:
            print(f"  {it + 1} reduces, {bad} wrong so far", flush=True)
    done, gave_up = reduce.status()
    results = [None] * world
    dist.all_gather_object(results, (bad, done, gave_up))
    if rank == 0:
        print(f"{world} ranks, {args.reduces} reduces each: wrong results per rank {[r[0] for r in results]}, "
              f"reduces completed {[r[1] for r in results]}, waits given up {[r[2] for r in results]}", flush=True)
    dist.barrier()
    reduce.close()
    sys.exit(1 if any(r[0] or r[2] for r in results) else 0)


main()
