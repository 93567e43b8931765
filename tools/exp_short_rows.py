#!/usr/bin/env python3
"""Where a long-tailed batch loses time: the same batch with its short rows cut to one token,
and with its long rows cut to one token.  (A row of one token costs a workgroup's fixed time only.)

    python tools/exp_short_rows.py [--batch 128] [--cut 256]
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import __graft_entry__  # noqa: E402
from bench import algorithmic_bytes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--cut", type=int, default=256)
    args = ap.parse_args()
    pkg = __graft_entry__.load_package()
    c = pkg._capi
    dev = torch.device("cuda:0")
    S, H, D, B, CTX = args.batch, 32, 128, 16, 2048
    base = [int(min(max(v, 1), CTX)) for v in np.random.default_rng(0).lognormal(math.log(300.0), 1.0, S)]
    bps = CTX // B
    NB = S * bps
    g = torch.Generator(device="cpu").manual_seed(0)
    copies = 3
    kcs = [torch.empty(NB, H, D // 8, B, 8, dtype=torch.float16, device=dev).normal_(0, 0.5) for _ in range(copies)]
    vcs = [torch.empty(NB, H, D, B, dtype=torch.float16, device=dev).normal_(0, 0.5) for _ in range(copies)]
    bt = torch.randperm(NB, generator=g).to(torch.int32).reshape(S, bps).to(dev)
    q = torch.empty(S, H, D, dtype=torch.float16, device=dev).normal_(0, 0.5)
    out = torch.empty_like(q)
    ws = torch.empty(max(c.decode_workspace_bytes(S, H, D, B, CTX), 8), dtype=torch.uint8, device=dev)
    variants = {
        "as is": base,
        f"rows <= {args.cut} cut to 1 token": [v if v > args.cut else 1 for v in base],
        f"rows > {args.cut} cut to 1 token": [v if v <= args.cut else 1 for v in base],
        "all rows 1 token": [1] * S,
    }
    print(f"batch {S}, long-tailed lengths (median {sorted(base)[S // 2]}, mean {sum(base) / S:.0f}, max {max(base)}); "
          f"{sum(1 for v in base if v <= args.cut)} rows of at most {args.cut} tokens")
    for name, ctxs in variants.items():
        cl = torch.tensor(ctxs, dtype=torch.int32, device=dev)
        total = sum(ctxs)
        c.single_query_cached_kv_attention_planned(out, q, kcs[0], vcs[0], D ** -0.5, bt, cl, B, CTX, ws, False, total)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for i in range(3):
                c.single_query_cached_kv_attention_planned(out, q, kcs[i % copies], vcs[i % copies], D ** -0.5, bt, cl, B, CTX, ws, True, total)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph):
            for i in range(9):
                c.single_query_cached_kv_attention_planned(out, q, kcs[i % copies], vcs[i % copies], D ** -0.5, bt, cl, B, CTX, ws, True, total)
        graph.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            graph.replay()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 9 * 1e3)
        ts.sort()
        nbytes = algorithmic_bytes(ctxs, H, D, B, 2)
        print(f"  {name:32s} {nbytes / 1e6:8.1f} MB  {ts[len(ts) // 2]:7.1f} us  {nbytes / ts[len(ts) // 2] / 1e3:6.0f} GB/s", flush=True)


main()
