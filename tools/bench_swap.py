#!/usr/bin/env python3
"""Times CacheEngine-style swaps of scattered blocks between the GPU cache and
pinned host memory: wall time of issuing + completing the copies of N blocks
for one layer's K and V (LLaMA-7B block = 128 KiB).  PA_SWAP_KERNEL_MIN_PAIRS=0
selects the one-memcpy-per-run path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__
pkg = __graft_entry__.load_package()
H, D, B = 32, 128, 16
NBG, NBC = 4096, 1024
gpu = [torch.randn(NBG, H, D // 8, B, 8, dtype=torch.float16, device="cuda:0") for _ in range(2)]
cpu = [torch.zeros(NBC, H, D // 8, B, 8, dtype=torch.float16).pin_memory() for _ in range(2)]
g = torch.Generator().manual_seed(0)
for n in (4, 32, 256, 1024):
    src = torch.randperm(NBG, generator=g)[:n].tolist()
    dst = torch.randperm(NBC, generator=g)[:n].tolist()
    out_map = dict(zip(src, dst))
    in_map = dict(zip(dst, src))
    for name, a, b, m in (("swap_out", gpu, cpu, out_map), ("swap_in", cpu, gpu, in_map)):
        for _ in range(2):
            for t in range(2): pkg.cache_ops.swap_blocks(a[t], b[t], m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(2): pkg.cache_ops.swap_blocks(a[t], b[t], m)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        nbytes = 2 * n * H * D * B * 2
        print(f"{name:8s} {n:5d} blocks x2: issue {t_issue * 1e3:7.2f} ms, done {t_all * 1e3:7.2f} ms, {nbytes / t_all / 1e9:6.1f} GB/s", flush=True)
    for s_, d_ in list(out_map.items())[:8]:
        assert torch.equal(cpu[0][d_], gpu[0][s_].cpu())
print("ok")
