#!/usr/bin/env python3
"""Times reshape_and_cache / gather_cached_kv / copy_blocks at decode-step and
prompt sizes (graph replay of 10 launches, rotating over several caches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__
pkg = __graft_entry__.load_package()
dev = "cuda:0"

def timeit(fn, n_group=10, groups=5):
    fn(0); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(n_group): fn(i)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n_group): fn(i)
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(groups):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n_group * 1e3)
    return sorted(ts)[len(ts) // 2]

def main():
    H, D, B = 32, 128, 16
    NB = 16384
    copies = 4
    kcs = [torch.zeros(NB, H, D // 8, B, 8, dtype=torch.float16, device=dev) for _ in range(copies)]
    vcs = [torch.zeros(NB, H, D, B, dtype=torch.float16, device=dev) for _ in range(copies)]
    for T in (256, 2560, 16384):
        qkv = torch.randn(T, 3, H, D, dtype=torch.float16, device=dev)
        if T == 256:
            slots = (torch.randperm(NB, device=dev)[:T] * B + 15).to(torch.int32)   # one token per block
        else:
            slots = torch.arange(T, device=dev, dtype=torch.int32) + 3 * B           # prompt: consecutive slots
        us = timeit(lambda i: pkg.cache_ops.reshape_and_cache(qkv[:, 1], qkv[:, 2], kcs[i % copies], vcs[i % copies], slots))
        moved = 2 * 2 * T * H * D * 2
        print(f"reshape_and_cache T={T:6d}: {us:8.1f} us  {moved / us / 1e3:8.1f} GB/s (algorithmic in+out)")
        out = torch.empty_like(qkv)
        us = timeit(lambda i: pkg.cache_ops.gather_cached_kv(out[:, 1], out[:, 2], kcs[i % copies], vcs[i % copies], slots))
        print(f"gather_cached_kv  T={T:6d}: {us:8.1f} us  {moved / us / 1e3:8.1f} GB/s")
    L = 4
    for pairs in (1, 8, 64):
        mapping = {i: [NB // 2 + i] for i in range(pairs)}
        us = timeit(lambda i: pkg.cache_ops.copy_blocks(kcs[:L], vcs[:L], mapping))
        moved = 2 * 2 * L * pairs * H * D * B * 2
        print(f"copy_blocks L={L} pairs={pairs:3d}: {us:8.1f} us  {moved / us / 1e3:8.1f} GB/s (read+write)")

main()
