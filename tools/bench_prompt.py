#!/usr/bin/env python3
"""Times prompt attention (multi_query_kv_attention) on LLaMA-shaped prompt
batches: graph replay of several launches, causal FLOPs = 4 * H * D * sum n(n+1)/2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import __graft_entry__
pkg = __graft_entry__.load_package()
dev = "cuda:0"

def timeit(fn, n_group=5, groups=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n_group): fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(groups):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n_group * 1e3)
    return sorted(ts)[len(ts) // 2]

def main():
    torch.manual_seed(0)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for dtype in (torch.float16, torch.bfloat16, torch.float32):
        if only and only not in str(dtype): continue
        for H, D, lens in ((32, 128, [2560]), (32, 128, [512] * 5), (32, 128, [2048] * 8), (32, 128, [4096] * 4),
                           (40, 128, [1024] * 16), (32, 128, [37, 211, 900, 1412]), (12, 64, [2048] * 8), (32, 128, [16384])):
            if dtype == torch.float32 and sum(lens) > 8192: continue
            T = sum(lens)
            qkv = torch.randn(T, 3, H, D, dtype=dtype, device=dev)
            q, k, v = qkv.unbind(dim=1)
            out = torch.empty(T, H, D, dtype=dtype, device=dev)
            scale = D ** -0.5
            us = timeit(lambda: pkg.attention_ops.multi_query_kv_attention(out, q, k, v, lens, scale))
            flops = 4 * H * D * sum(n * (n + 1) // 2 for n in lens)
            line = f"{str(dtype)[6:]:9s} H={H} D={D} lens={len(lens)}x{max(lens):5d} T={T:6d}: {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s"
            if dtype != torch.float32 and len(set(lens)) == 1:
                n = lens[0]
                q4 = q.reshape(len(lens), n, H, D).transpose(1, 2)
                k4 = k.reshape(len(lens), n, H, D).transpose(1, 2)
                v4 = v.reshape(len(lens), n, H, D).transpose(1, 2)
                try:
                    us_t = timeit(lambda: F.scaled_dot_product_attention(q4, k4, v4, is_causal=True, scale=scale))
                    line += f"   (torch SDPA {us_t:9.1f} us, {flops / us_t / 1e6:6.1f} TFLOP/s)"
                except Exception as e:
                    line += f"   (torch SDPA failed: {type(e).__name__})"
            print(line, flush=True)

main()
