#!/usr/bin/env python3
"""Small decode batches through BOTH entry points, eager and from a HIP graph.

The reference's caller invokes attention_ops.single_query_cached_kv_attention(out, query, key_cache,
value_cache, scale, block_tables, context_lens, block_size, max_context_len) once per layer
(vllm/model_executor/layers/attention.py:81-91): that call sizes its scratch and plans its launch by itself.
This package's own layer calls the _planned form with a workspace and plan shared by the layers of a step.
benchmarks/benchmark_latency.py:36-57 of the reference runs batch 8 by default.

    python tools/sweep_small_batch.py [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import __graft_entry__  # noqa: E402
import sweep_decode  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--head-size", type=int, default=128)
    args = ap.parse_args()
    pkg = __graft_entry__.load_package()
    device = torch.device("cuda:0")
    rows = []
    print(f"{args.heads} heads x {args.head_size}, block 16, fp16; us per launch (mean of groups of 10)")
    print(f"{'batch':>5s} {'ctx':>5s} | {'reference signature':>21s} | {'planned (shared plan)':>21s} | difference (graph)")
    print(f"{'':>5s} {'':>5s} | {'eager':>10s} {'graph':>10s} | {'eager':>10s} {'graph':>10s} |")
    for ctx in (128, 1024):
        for S in (1, 2, 4, 8, 16):
            cfg = dict(name=f"bs{S}_ctx{ctx}", S=S, H=args.heads, D=args.head_size, B=16, ctx=ctx, dt="f16", ragged=False, share=1)
            res = {}
            for planned in (False, True):
                for graph in (False, True):
                    sweep_decode.PLANNED, sweep_decode.USE_GRAPH = planned, graph
                    r = sweep_decode.run(cfg, pkg, device, launches=60)
                    res[("planned" if planned else "reference", "graph" if graph else "eager")] = r["us_mean"]
            row = dict(batch=S, ctx=ctx, mbytes=r["mbytes"], **{f"{k[0]}_{k[1]}_us": v for k, v in res.items()})
            rows.append(row)
            print(f"{S:5d} {ctx:5d} | {res[('reference', 'eager')]:10.2f} {res[('reference', 'graph')]:10.2f} | "
                  f"{res[('planned', 'eager')]:10.2f} {res[('planned', 'graph')]:10.2f} | "
                  f"{res[('reference', 'graph')] - res[('planned', 'graph')]:+6.2f} us   "
                  f"({100 * r['mbytes'] * 1e6 / (res[('reference', 'graph')] * 1e-6) / 8e12:4.1f} % of 8 TB/s through the reference signature)",
                  flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
