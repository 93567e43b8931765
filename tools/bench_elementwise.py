#!/usr/bin/env python3
"""Times rotary_embedding_neox / rms_norm / silu_and_mul at LLaMA-7B/13B
decode-step (256 tokens) and prompt (2048, 16384 tokens) sizes: graph replay of
10 launches rotating over several buffers, algorithmic GB/s (read + write)."""
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
    copies = 10   # every launch of a group touches its own buffers (no cache reuse between launches)
    dt = torch.float16
    for hidden, inter, H, D in ((4096, 11008, 32, 128), (5120, 13824, 40, 128)):
        for T in (256, 2048, 16384):
            xs = [torch.randn(T, hidden, dtype=dt, device=dev) for _ in range(copies)]
            outs = [torch.empty_like(x) for x in xs]
            w = torch.ones(hidden, dtype=dt, device=dev)
            us = timeit(lambda i: pkg.layernorm_ops.rms_norm(outs[i % copies], xs[i % copies], w, 1e-6))
            print(f"rms_norm      hidden={hidden} T={T:6d}: {us:8.1f} us  {2 * T * hidden * 2 / us / 1e3:8.1f} GB/s")
            c2 = copies if T < 16384 else 4
            gus = [torch.randn(T, 2 * inter, dtype=dt, device=dev) for _ in range(c2)]
            ys = [torch.empty(T, inter, dtype=dt, device=dev) for _ in range(c2)]
            us = timeit(lambda i: pkg.activation_ops.silu_and_mul(ys[i % c2], gus[i % c2]))
            print(f"silu_and_mul  d={inter:6d}    T={T:6d}: {us:8.1f} us  {3 * T * inter * 2 / us / 1e3:8.1f} GB/s")
            del gus, ys
            qkvs = [torch.randn(T, 3 * H * D, dtype=dt, device=dev) for _ in range(copies)]
            pos = torch.randint(0, 2048, (T,), device=dev)
            table = torch.randn(2048, D, dtype=dt, device=dev)
            def rot(i):
                q, k, _ = qkvs[i % copies].chunk(3, dim=-1)
                pkg.pos_encoding_ops.rotary_embedding_neox(pos, q, k, D, table)
            us = timeit(rot)
            print(f"rotary_neox   heads={H}     T={T:6d}: {us:8.1f} us  {2 * 2 * T * H * D * 2 / us / 1e3:8.1f} GB/s")
            del qkvs, xs, outs

main()
