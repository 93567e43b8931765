#!/usr/bin/env python3
"""A decode step's cache write + attention per layer, as two launches (reshape_and_cache, then
single_query_cached_kv_attention on a shared plan) and as one (the append form), on one GPU.

    python tools/bench_append_decode.py [--batch 256] [--heads 32] [--head-size 128] [--ctx 1024] [--layers 8]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--head-size", type=int, default=128)
    ap.add_argument("--block-size", type=int, default=16)
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", default="float16")
    ap.add_argument("--ragged", action="store_true")
    args = ap.parse_args()
    pkg = __graft_entry__.load_package()
    c = pkg._capi
    dev = torch.device("cuda:0")
    dt = getattr(torch, args.dtype)
    S, H, D, B, CTX, L = args.batch, args.heads, args.head_size, args.block_size, args.ctx, args.layers
    x = 16 // torch.tensor([], dtype=dt).element_size()
    bps = (CTX + B - 1) // B
    NB = S * bps
    g = torch.Generator(device="cpu").manual_seed(0)
    caches = [(torch.empty(NB, H, D // x, B, x, dtype=dt, device=dev).normal_(0, 0.5),
               torch.empty(NB, H, D, B, dtype=dt, device=dev).normal_(0, 0.5)) for _ in range(L)]
    bt = torch.randperm(NB, generator=g).to(torch.int32).reshape(S, bps).to(dev)
    ctxs = torch.randint(1, CTX + 1, (S,), generator=g) if args.ragged else torch.full((S,), CTX)
    cl = ctxs.to(torch.int32).to(dev)
    total = int(ctxs.sum())
    pos = (ctxs - 1).to(dev)
    slots = (bt[torch.arange(S, device=dev), pos // B].long() * B + pos % B).to(torch.int32)
    qkv = torch.empty(S, 3, H, D, dtype=dt, device=dev).normal_(0, 0.5)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
    out = torch.empty(S, H, D, dtype=dt, device=dev)
    ws = torch.empty(max(c.decode_workspace_bytes(S, H, D, B, CTX), 8), dtype=torch.uint8, device=dev)
    scale = D ** -0.5

    def two_calls():
        for i, (kc, vc) in enumerate(caches):
            c.reshape_and_cache(k, v, kc, vc, slots)
            c.single_query_cached_kv_attention_planned(out, q, kc, vc, scale, bt, cl, B, CTX, ws, i > 0, total)

    def one_call():
        for i, (kc, vc) in enumerate(caches):
            c.append_single_query_cached_kv_attention(out, q, k, v, kc, vc, scale, bt, cl, B, CTX, ws, 1 if i else 0, total)

    def decode_only():
        for i, (kc, vc) in enumerate(caches):
            c.single_query_cached_kv_attention_planned(out, q, kc, vc, scale, bt, cl, B, CTX, ws, i > 0, total)

    def time_it(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.steps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.steps / L * 1e3

    print(f"batch {S} x {H} heads x {D}, block {B}, ctx {'<=' if args.ragged else ''}{CTX}, {args.dtype}, {L} layers; us per layer")
    for name, fn in (("attention only", decode_only), ("reshape_and_cache + attention", two_calls),
                     ("append form (one launch)", one_call), ("reshape_and_cache + attention", two_calls),
                     ("append form (one launch)", one_call), ("attention only", decode_only)):
        print(f"  {name:32s} {time_it(fn):9.2f}", flush=True)


main()
