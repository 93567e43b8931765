#!/usr/bin/env python3
"""Does the relative placement of a layer's key and value caches matter to the decode kernel?
Both caches are carved from one slab; the value cache starts `offset` bytes after the key cache ends.

    python tools/exp_cache_placement.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__  # noqa: E402
from bench import algorithmic_bytes  # noqa: E402


def main():
    pkg = __graft_entry__.load_package()
    c = pkg._capi
    dev = torch.device("cuda:0")
    S, H, D, B, CTX = [int(x) for x in os.environ.get("SHAPE", "256,32,128,16,1024").split(",")]
    bps = CTX // B
    NB = S * bps
    n = NB * H * D * B  # elements per cache
    g = torch.Generator(device="cpu").manual_seed(0)
    bt = torch.randperm(NB, generator=g).to(torch.int32).reshape(S, bps).to(dev)
    cl = torch.full((S,), CTX, dtype=torch.int32, device=dev)
    q = torch.empty(S, H, D, dtype=torch.float16, device=dev).normal_(0, 0.5)
    out = torch.empty_like(q)
    ws = torch.empty(max(c.decode_workspace_bytes(S, H, D, B, CTX), 8), dtype=torch.uint8, device=dev)
    nb = algorithmic_bytes([CTX] * S, H, D, B, 2)
    copies = 3 if n * 4 > (1 << 30) else 8   # small working sets: more copies, so that the Infinity Cache cannot serve them
    pad = 1 << 21
    slabs = [torch.empty(2 * n + pad // 2, dtype=torch.float16, device=dev).normal_(0, 0.5) for _ in range(copies)]
    offsets = [int(x) for x in os.environ.get("OFFSETS", "0,128,256,512,1024,2048,4096,8192,16384,65536,262144,1048576,0").split(",")]
    for rnd in range(int(os.environ.get("ROUNDS", "2"))):
        for off in offsets:
            caches = []
            for s in slabs:
                kc = s[:n].view(NB, H, D // 8, B, 8)
                vc = s[n + off // 2:2 * n + off // 2].view(NB, H, D, B)
                caches.append((kc, vc))
            c.single_query_cached_kv_attention_planned(out, q, caches[0][0], caches[0][1], D ** -0.5, bt, cl, B, CTX, ws, False, S * CTX)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(6):
                kc, vc = caches[i % copies]
                c.single_query_cached_kv_attention_planned(out, q, kc, vc, D ** -0.5, bt, cl, B, CTX, ws, True, S * CTX)
            a.record()
            for i in range(30):
                kc, vc = caches[i % copies]
                c.single_query_cached_kv_attention_planned(out, q, kc, vc, D ** -0.5, bt, cl, B, CTX, ws, True, S * CTX)
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / 30 * 1e3
            print(f"round {rnd}: value cache {off:8d} B after the key cache: {us:7.1f} us  {nb / us / 1e3:6.0f} GB/s", flush=True)


main()
