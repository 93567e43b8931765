#!/usr/bin/env python3
"""Experiment: how much of a ragged batch's decode time is per-XCD imbalance?

The decode kernel deals groups of G sequences round-robin to the 8 XCDs, so
the batch order fixes each XCD's share of the bytes.  The same multiset of
context lengths is timed in several orders: as drawn, sorted, arranged so that
the XCDs' sums are equal (greedy longest-first assignment), and arranged so
that they are as unequal as possible.

    python tools/exp_ragged_balance.py [--ctx 1024] [--batch 256]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__  # noqa: E402
from bench import algorithmic_bytes  # noqa: E402


def xcd_of_position(pos, G):
    return (pos // G) % 8


def arrange(ctxs, G, balanced):
    """Order the sequences so that XCD sums are equal (balanced) or extreme."""
    S = len(ctxs)
    slots = {x: [p for p in range(S) if xcd_of_position(p, G) == x] for x in range(8)}
    order = sorted(range(S), key=lambda i: -ctxs[i])
    out = [None] * S
    if balanced:
        sums = [0] * 8
        fill = [0] * 8
        for i in order:
            x = min((x for x in range(8) if fill[x] < len(slots[x])), key=lambda x: sums[x])
            out[slots[x][fill[x]]] = ctxs[i]
            fill[x] += 1
            sums[x] += ctxs[i]
    else:
        k = 0
        for x in range(8):
            for p in slots[x]:
                out[p] = ctxs[order[k]]
                k += 1
    return out


def arrange_two_level(ctxs, G, U=4):
    """Units of U consecutive sequences dealt to the XCDs in a snake by unit
    weight; inside an XCD, its sequences longest first."""
    S = len(ctxs)
    slots = {x: [p for p in range(S) if xcd_of_position(p, G) == x] for x in range(8)}
    units = [list(range(u, min(u + U, S))) for u in range(0, S, U)]
    order = sorted(range(len(units)), key=lambda u: -sum(ctxs[i] for i in units[u]))
    mine = {x: [] for x in range(8)}
    for r, u in enumerate(order):
        i, j = divmod(r, 8)
        mine[j if i % 2 == 0 else 7 - j] += units[u]
    out = [None] * S
    spill = []
    for x in range(8):
        seqs = sorted(mine[x], key=lambda i: -ctxs[i])
        for p, i in zip(slots[x], seqs):
            out[p] = ctxs[i]
        spill += [ctxs[i] for i in seqs[len(slots[x]):]]
    for p in range(S):
        if out[p] is None:
            out[p] = spill.pop()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lognormal", action="store_true")
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--heads", type=int, default=32)
    args = ap.parse_args()
    pkg = __graft_entry__.load_package()
    dev = torch.device("cuda:0")
    S, H, D, B, CTX = args.batch, args.heads, 128, 16, args.ctx
    G = max(1, min(16, 128 // H))
    g = torch.Generator(device="cpu").manual_seed(0)
    drawn = torch.randint(1, CTX + 1, (S,), generator=g).tolist()
    if args.lognormal:
        import numpy as np
        drawn = [int(min(max(v, 1), CTX)) for v in np.random.default_rng(0).lognormal(np.log(300.0), 1.0, S)]
    bps = (CTX + B - 1) // B
    NB = S * bps
    kc = torch.empty(NB, H, D // 8, B, 8, dtype=torch.float16, device=dev).normal_(0, 0.5)
    vc = torch.empty(NB, H, D, B, dtype=torch.float16, device=dev).normal_(0, 0.5)
    bt = torch.randperm(NB, generator=g).to(torch.int32).reshape(S, bps).to(dev)
    q = torch.empty(S, H, D, dtype=torch.float16, device=dev).normal_(0, 0.5)
    out = torch.empty_like(q)
    op = pkg.attention_ops.single_query_cached_kv_attention
    orders = {
        "as drawn": drawn,
        "sorted descending": sorted(drawn, reverse=True),
        "XCD sums balanced": arrange(drawn, G, True),
        "XCD sums extreme": arrange(drawn, G, False),
        "units of 4 + LPT in XCD": arrange_two_level(drawn, G),
        f"uniform {sum(drawn) // S}": [sum(drawn) // S] * S,
    }
    for name, ctxs in orders.items():
        cl = torch.tensor(ctxs, dtype=torch.int32, device=dev)
        sums = [sum(c for p, c in enumerate(ctxs) if xcd_of_position(p, G) == x) for x in range(8)]
        for _ in range(3):
            op(out, q, kc, vc, D ** -0.5, bt, cl, B, max(ctxs))
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                op(out, q, kc, vc, D ** -0.5, bt, cl, B, max(ctxs))
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 10 * 1e3)
        us = sorted(ts)[len(ts) // 2]
        nbytes = algorithmic_bytes(ctxs, H, D, B, 2)
        print(f"{name:22s} {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s   XCD token sums: max/mean = "
              f"{max(sums) / (sum(sums) / 8):.3f}", flush=True)


main()
