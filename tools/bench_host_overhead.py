#!/usr/bin/env python3
"""Host-side cost of one call of each operator module function (tiny inputs,
many calls, wall time per call with the queue kept short): what an eager
decoder loop pays per layer on the CPU besides the kernels themselves."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__
pkg = __graft_entry__.load_package()
dev = "cuda:0"
S, H, D, B, NB = 4, 8, 128, 16, 64
dt = torch.float16
qkv = torch.randn(S, 3, H, D, dtype=dt, device=dev)
q, k, v = qkv.unbind(1)
out = torch.empty(S, H, D, dtype=dt, device=dev)
kc = torch.zeros(NB, H, D // 8, B, 8, dtype=dt, device=dev)
vc = torch.zeros(NB, H, D, B, dtype=dt, device=dev)
bt = torch.arange(S * 4, dtype=torch.int32, device=dev).reshape(S, 4)
cl = torch.full((S,), 40, dtype=torch.int32, device=dev)
slots = torch.arange(S, dtype=torch.int32, device=dev) * B
x = torch.randn(S, H * D, dtype=dt, device=dev)
w = torch.ones(H * D, dtype=dt, device=dev)
y = torch.empty_like(x)
gu = torch.randn(S, 2 * 512, dtype=dt, device=dev)
act = torch.empty(S, 512, dtype=dt, device=dev)
pos = torch.arange(S, device=dev)
table = torch.randn(64, D, dtype=dt, device=dev)
q2, k2 = x.clone(), x.clone()
ops = {
    "single_query_cached_kv_attention": lambda: pkg.attention_ops.single_query_cached_kv_attention(out, q, kc, vc, 0.1, bt, cl, B, 64),
    "multi_query_kv_attention": lambda: pkg.attention_ops.multi_query_kv_attention(out, q, k, v, [S], 0.1),
    "reshape_and_cache": lambda: pkg.cache_ops.reshape_and_cache(k, v, kc, vc, slots),
    "copy_blocks (2 layers, 1 pair)": lambda: pkg.cache_ops.copy_blocks([kc, kc], [vc, vc], {1: [2]}),
    "rotary_embedding_neox": lambda: pkg.pos_encoding_ops.rotary_embedding_neox(pos, q2, k2, D, table),
    "rms_norm": lambda: pkg.layernorm_ops.rms_norm(y, x, w, 1e-6),
    "silu_and_mul": lambda: pkg.activation_ops.silu_and_mul(act, gu),
    "torch add (for scale)": lambda: torch.add(x, x, out=y),
}
for name, fn in ops.items():
    for _ in range(50): fn()
    torch.cuda.synchronize()
    n, t = 0, 0.0
    for _ in range(20):
        t0 = time.perf_counter()
        for _ in range(100): fn()
        t += time.perf_counter() - t0
        n += 100
        torch.cuda.synchronize()
    print(f"{name:36s} {t / n * 1e6:6.2f} us per call on the host")
