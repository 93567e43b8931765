#!/usr/bin/env python3
"""rotary_embedding_neox + reshape_and_cache for a decode step's tokens: the
reference's two calls against the one-pass operator, HIP-graph replay of 32
layers' worth of calls (one layer's caches each, so nothing is served from the
Infinity Cache by accident), LLaMA-7B sizes.

    python tools/bench_rotary_cache.py [--batch 256]
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
    ap.add_argument("--layers", type=int, default=32)
    args = ap.parse_args()
    pkg = __graft_entry__.load_package()
    dev, dt = torch.device("cuda:0"), torch.float16
    S, H, D, B, L, CTX = args.batch, args.heads, args.head_size, 16, args.layers, 1024
    bps = CTX // B
    NB = S * bps
    g = torch.Generator(device="cpu").manual_seed(0)
    bt = torch.randperm(NB, generator=g).to(torch.int32).reshape(S, bps).to(dev)
    slots = (bt[:, -1].long() * B + (CTX - 1) % B).to(torch.int32)
    pos = torch.full((S,), CTX - 1, dtype=torch.long, device=dev)
    kcs = [torch.empty(NB, H, D // 8, B, 8, dtype=dt, device=dev).normal_() for _ in range(L)]
    vcs = [torch.empty(NB, H, D, B, dtype=dt, device=dev).normal_() for _ in range(L)]
    qkv = torch.empty(S, 3, H, D, dtype=dt, device=dev).normal_()
    q, k, v = qkv.unbind(dim=1)
    q2, k2 = qkv.view(S, 3, H * D)[:, 0], qkv.view(S, 3, H * D)[:, 1]
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(torch.arange(2048).float(), inv)
    cs = torch.cat((fr.cos(), fr.sin()), dim=-1).to(dt).to(dev)

    def two_calls():
        for l in range(L):
            pkg.pos_encoding_ops.rotary_embedding_neox(pos, q2, k2, D, cs)
            pkg.cache_ops.reshape_and_cache(k, v, kcs[l], vcs[l], slots)

    def one_pass():
        for l in range(L):
            pkg.cache_ops.rotary_embedding_neox_reshape_and_cache(pos, q, k, v, kcs[l], vcs[l], slots, cs)

    def only_reshape():
        for l in range(L):
            pkg.cache_ops.reshape_and_cache(k, v, kcs[l], vcs[l], slots)

    print(f"batch {S}, {H} heads x {D}, fp16, {L} layers per replay; us per layer")
    for name, fn in (("rotary_embedding_neox, then reshape_and_cache", two_calls),
                     ("one pass", one_pass), ("reshape_and_cache alone", only_reshape)):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fn()
        graph.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            graph.replay()
        b.record()
        torch.cuda.synchronize()
        print(f"  {name:48s} {a.elapsed_time(b) / 10 / L * 1e3:7.2f}")


main()
