#!/usr/bin/env python3
"""Sweeps decode-attention shapes on one GPU and prints achieved HBM GB/s
(algorithmic bytes / mean launch time; for batches whose sequences share
physical blocks, the distinct bytes).  Small working sets are rotated over
several cache copies so that the 256 MiB Infinity Cache cannot serve them.

    python tools/sweep_decode.py [--only NAME_SUBSTR] [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import __graft_entry__  # noqa: E402
from bench import algorithmic_bytes  # noqa: E402

USE_GRAPH = False
PLANNED = False  # one launch plan for all timed calls (the layers of a step), not one per call
SEQUENTIAL = False  # physical block i = logical block i (best case for DRAM page locality)
DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


def configs():
    c = []
    def add(name, S=256, H=32, D=128, B=16, ctx=1024, dt="f16", ragged=False, share=1):
        c.append(dict(name=name, S=S, H=H, D=D, B=B, ctx=ctx, dt=dt, ragged=ragged, share=share))
    add("north_star")
    add("c2_opt125m", H=12, D=64)
    for ctx in (128, 512, 2048, 4096):
        add(f"ns_ctx{ctx}", ctx=ctx)
    for S in (1, 4, 16, 64, 128):
        add(f"ns_bs{S}", S=S)
    for S in (2, 6, 7, 8, 12):  # around the point where the rows alone fill the 256 CUs (32 heads: 8 sequences)
        add(f"ns_bs{S}", S=S)
    add("ns_bs6_ctx512", S=6, ctx=512)
    add("c2_bs16", S=16, H=12, D=64)
    add("tiny_bs1_ctx64", S=1, ctx=64)
    add("tiny_bs4_ctx128", S=4, ctx=128)
    add("tiny_bs4_ctx512", S=4, ctx=512)
    add("tiny_bs16_ctx128", S=16, ctx=128)
    add("long_bs1_ctx16k", S=1, ctx=16384)
    add("long_bs4_ctx8k", S=4, ctx=8192)
    for B in (8, 32):
        add(f"ns_block{B}", B=B)
    for D in (64, 80, 96):
        add(f"ns_head{D}", D=D)
    add("ns_ragged", ragged=True)
    add("ns_ragged_sorted", ragged="sorted")
    add("ns_ragged_lognormal", ragged="lognormal", ctx=2048)
    add("ns_ragged_lognormal_bs128", ragged="lognormal", ctx=2048, S=128)
    add("ns_ragged_lognormal_1k", ragged="lognormal", ctx=1024)
    add("ns_outlier_8k", ragged="outlier", ctx=8192)
    add("ns_outlier_2k_bs64", ragged="outlier", ctx=2048, S=64)
    add("ns_bs64_ctx2048", ctx=2048, S=64)
    add("ns_bs32_ctx2048", ctx=2048, S=32)
    add("ns_bs16_ctx4096", ctx=4096, S=16)
    add("ns_bs40_ctx4096", ctx=4096, S=40)
    add("ns_bs32_ragged2048", ctx=2048, S=32, ragged=True)
    add("ns_bs64_ragged2048", ctx=2048, S=64, ragged=True)
    add("ns_bs256_ragged2048", ctx=2048, S=256, ragged=True)
    add("ns_bf16", dt="bf16")
    add("ns_f32", dt="f32", S=128)
    add("c4_llama13b_tp2", H=20)
    add("c5_neox20b_tp4_n3", H=16, D=96, share=3)
    add("tp8_llama65b", H=8)
    add("opt30b_h56", H=56)
    add("h64", H=64, S=128)
    add("h1", H=1, S=256, ctx=2048)
    add("bs100_h40", H=40, S=100)
    add("opt2.7b_h32_d80_bs37", H=32, D=80, S=37)
    add("c2_block8", H=12, D=64, B=8)
    add("c2_ragged", H=12, D=64, ragged=True)
    return c


def run(cfg, pkg, device, launches=30):
    S, H, D, B, ctx = cfg["S"], cfg["H"], cfg["D"], cfg["B"], cfg["ctx"]
    dt = DT[cfg["dt"]]
    es = torch.tensor([], dtype=dt).element_size()
    x = 16 // es
    g = torch.Generator(device="cpu")
    g.manual_seed(0)
    if cfg["ragged"]:
        ctxs = torch.randint(1, ctx + 1, (S,), generator=g).tolist()
        if cfg["ragged"] == "sorted":
            ctxs.sort(reverse=True)
        elif cfg["ragged"] == "lognormal":  # long-tailed, like the lengths of a chat corpus
            rs = __import__("numpy").random.default_rng(0)
            ctxs = [int(min(max(v, 1), ctx)) for v in rs.lognormal(__import__("math").log(300.0), 1.0, S)]
        elif cfg["ragged"] == "outlier":   # one long sequence among short ones
            ctxs = [ctx // 32] * S
            ctxs[S // 2] = ctx
    else:
        ctxs = [ctx] * S
    bps = (ctx + B - 1) // B
    share = cfg["share"]
    # sequences in a group of `share` use the same physical blocks for the
    # first half of their context (a shared prompt), distinct ones after
    groups = (S + share - 1) // share
    half = bps // 2 if share > 1 else 0
    NB = groups * half + S * (bps - half)
    tile_bytes = H * D * B * es
    ws = 2 * NB * tile_bytes
    copies = max(1, min(8, int((1.5 * 2**30) // max(ws, 1)) + 1)) if ws < 2**30 else 1
    kcs = [torch.empty(NB, H, D // x, B, x, dtype=dt, device=device).normal_(0, 0.5) for _ in range(copies)]
    vcs = [torch.empty(NB, H, D, B, dtype=dt, device=device).normal_(0, 0.5) for _ in range(copies)]
    perm = torch.randperm(NB, generator=g).to(torch.int32)
    if SEQUENTIAL:
        perm = torch.arange(NB, dtype=torch.int32)
    bt = torch.zeros(S, bps, dtype=torch.int32)
    shared = perm[:groups * half].reshape(groups, half) if half else None
    own = perm[groups * half:].reshape(S, bps - half)
    for s in range(S):
        if half:
            bt[s, :half] = shared[s // share]
        bt[s, half:] = own[s]
    bt = bt.to(device)
    cl = torch.tensor(ctxs, dtype=torch.int32, device=device)
    qkv = torch.empty(S, 3, H, D, dtype=dt, device=device).normal_(0, 0.5)
    q = qkv[:, 0]
    out = torch.empty(S, H, D, dtype=dt, device=device)
    scale = 1.0 / D ** 0.5
    op = pkg.attention_ops.single_query_cached_kv_attention
    if PLANNED:
        ws = torch.empty(pkg.attention_ops.decode_workspace_bytes(S, H, D, B, ctx), dtype=torch.uint8, device=device)
        planned = pkg.attention_ops.single_query_cached_kv_attention_planned
        total = -1 if os.environ.get("SWEEP_NO_HINT") else sum(ctxs)  # the host-side sum of context_lens
        planned(out, q, kcs[0], vcs[0], scale, bt, cl, B, ctx, ws, False, total)

        def op(out, q, kc, vc, scale, bt, cl, B, ctx):  # noqa: F811
            planned(out, q, kc, vc, scale, bt, cl, B, ctx, ws, True, total)
    for i in range(3):
        op(out, q, kcs[i % copies], vcs[i % copies], scale, bt, cl, B, ctx)
    torch.cuda.synchronize()
    # groups of back-to-back launches between one event pair: for short
    # kernels a per-launch event pair would mostly time the events.  With
    # --graph the group is captured into a HIP graph and replayed, which takes
    # the host's launch cost out of the measurement.
    per_group = 10
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(max(launches // per_group, 3))]
    def group(i0):
        for k in range(per_group):
            op(out, q, kcs[(i0 + k) % copies], vcs[(i0 + k) % copies], scale, bt, cl, B, ctx)
    if USE_GRAPH:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            group(0)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            group(0)
        graph.replay()
        torch.cuda.synchronize()
        for a, b in evs:
            a.record()
            graph.replay()
            b.record()
    else:
        i = 0
        for a, b in evs:
            a.record()
            group(i)
            i += per_group
            b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) / per_group for a, b in evs)
    mean = sum(ts) / len(ts)
    nbytes = algorithmic_bytes(ctxs, H, D, B, es)
    if half:
        # Shared-prefix batch: the algorithmic figure counts a shared block once per reader, but
        # only its first reader brings it from HBM.  The roofline fraction is priced on the bytes
        # that are distinct: shared tokens once per group, own tokens once per sequence.
        shared_tokens = groups * half * B
        own_tokens = sum(max(c - half * B, 0) for c in ctxs)
        nbytes_unique = nbytes - sum(ctxs) * H * 2 * D * es + (shared_tokens + own_tokens) * H * 2 * D * es
    else:
        nbytes_unique = nbytes
    del kcs, vcs
    torch.cuda.empty_cache()
    return dict(cfg, us_mean=mean * 1e3, us_min=ts[0] * 1e3, gbps=nbytes_unique / (mean * 1e-3) / 1e9,
                mbytes=nbytes_unique / 1e6, copies=copies, gbps_per_reader=nbytes / (mean * 1e-3) / 1e9)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--names", default="", help="comma-separated exact configuration names")
    ap.add_argument("--json", default="")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--sequential", action="store_true")
    ap.add_argument("--planned", action="store_true")
    args = ap.parse_args()
    global USE_GRAPH, SEQUENTIAL, PLANNED
    PLANNED = args.planned
    USE_GRAPH = args.graph
    SEQUENTIAL = args.sequential
    pkg = __graft_entry__.load_package()
    device = torch.device("cuda:0")
    rows = []
    for cfg in configs():
        if args.only and args.only not in cfg["name"]:
            continue
        if args.names and cfg["name"] not in args.names.split(","):
            continue
        r = run(cfg, pkg, device)
        rows.append(r)
        print(f"{r['name']:24s} S={r['S']:<4d} H={r['H']:<3d} D={r['D']:<4d} B={r['B']:<3d} ctx={r['ctx']:<5d} "
              f"{r['dt']:4s} {r['mbytes']:9.1f} MB  {r['us_mean']:9.1f} us (min {r['us_min']:8.1f})  "
              f"{r['gbps']:7.0f} GB/s  {100 * r['gbps'] / 8000:5.1f}% of 8 TB/s"
              + (f"  (distinct bytes; {r['gbps_per_reader']:.0f} GB/s counted per reader)" if r["share"] > 1 else ""),
              flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
