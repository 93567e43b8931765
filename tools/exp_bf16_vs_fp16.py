#!/usr/bin/env python3
"""Is the bf16 decode launch slower than the fp16 one because of its code or because of its data?
Times the north-star shape four ways on the same cache memory: bf16 kernel on bf16 normal data,
fp16 kernel on the very same bytes (reinterpreted), fp16 kernel on fp16 normal data, bf16 kernel
on those bytes.

    python tools/exp_bf16_vs_fp16.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__  # noqa: E402


def main():
    pkg = __graft_entry__.load_package()
    c = pkg._capi
    from vllm_amd.worker import carve_kv_cache, kv_slab_elements
    dev = torch.device("cuda:0")
    S, H, D, B, CTX, L = 256, 32, 128, 16, 1024, 6
    bps = CTX // B
    NB = S * bps
    g = torch.Generator(device="cpu").manual_seed(0)
    bt = torch.randperm(NB, generator=g).to(torch.int32).reshape(S, bps).to(dev)
    cl = torch.full((S,), CTX, dtype=torch.int32, device=dev)
    slab = torch.empty(kv_slab_elements(L, NB, (H, D, B), 2), dtype=torch.float16, device=dev)
    caches = [carve_kv_cache(slab, l, NB, (H, D // 8, B, 8), (H, D, B)) for l in range(L)]
    ws = torch.empty(max(c.decode_workspace_bytes(S, H, D, B, CTX), 8), dtype=torch.uint8, device=dev)

    def fill(dtype):
        for kc, vc in caches:
            kc.view(dtype).normal_(0, 0.5)
            vc.view(dtype).normal_(0, 0.5)

    def run(dtype, n=60):
        q = torch.empty(S, H, D, dtype=dtype, device=dev).normal_(0, 0.5)
        out = torch.empty_like(q)
        kc0, vc0 = caches[0]
        c.single_query_cached_kv_attention_planned(out, q, kc0.view(dtype), vc0.view(dtype), D ** -0.5, bt, cl, B, CTX, ws, False, S * CTX)
        ts = []
        for i in range(n):
            kc, vc = caches[i % L]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            c.single_query_cached_kv_attention_planned(out, q, kc.view(dtype), vc.view(dtype), D ** -0.5, bt, cl, B, CTX, ws, True, S * CTX)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts = sorted(ts[10:])
        return ts[len(ts) // 2]

    for data in (torch.bfloat16, torch.float16):
        fill(data)
        for kern in (torch.bfloat16, torch.float16, torch.bfloat16, torch.float16):
            print(f"data written as {str(data)[6:]:9s} kernel {str(kern)[6:]:9s}: {run(kern):7.1f} us", flush=True)


main()
