#!/usr/bin/env python3
"""Benchmark of the PagedAttention decode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): output tokens/s + paged-attention HBM GB/s, LLaMA-7B
TP=1, batch 256.  One STEP is one decode iteration of the hot path over the
whole model: for each of the 32 layers, `cache_ops.reshape_and_cache` writes
the batch's new K/V vectors into that layer's paged cache and
`attention_ops.single_query_cached_kv_attention_planned` attends over it
(batch 256, 32 heads x 128, block 16, context 1024, fp16; every logical block a
distinct physical block; one 2 GiB K and one 2 GiB V cache per layer, all
resident in HBM before timing starts).  The attention's launch plan (the order
in which rows are dealt to the XCDs) depends on the context lengths only and is
made once per step, by layer 0, as in this package's PagedAttention layer.  value = tokens produced per second by that path =
batch * steps / elapsed.  The GEMMs, sampler and scheduler of the full model
are outside this path and are not run.

With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank
runs the same per-GPU workload on its own caches: the (sequence, head) units
are independent, so there is no data-path collective; value aggregates all
ranks (weak scaling).  Timing is barrier + synchronize on both sides, MAX over
ranks.

The JSON line also carries
  roofline      achieved HBM GB/s of the decode kernel: algorithmic bytes per
                launch / mean launch time measured with HIP events on the
                launch stream, against the 8 TB/s HBM3E peak;
  cpu_baseline  the C oracle (a CPU port of the reference algorithm) timed on
                this host for one layer of the same workload, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak

WORKLOADS = {
    # name: (num_heads, head_size, num_layers, description)
    "llama7b": (32, 128, 32, "LLaMA-7B decode hot path"),
    "opt125m_c2": (12, 64, 12, "microbench shape of BASELINE.json configs[1] (12 heads x 64)"),
}


def algorithmic_bytes(ctx_lens, num_heads, head_size, block_size, elem_size):
    """Bytes one decode launch must move: each context token's K and V read
    once, q read and out written once, the used block-table entries and the
    context lengths read once."""
    S = len(ctx_lens)
    kv = sum(num_heads * 2 * c * head_size * elem_size for c in ctx_lens)
    qo = 2 * S * num_heads * head_size * elem_size
    tbl = sum((c + block_size - 1) // block_size * 4 for c in ctx_lens) + 4 * S
    return kv + qo + tbl


def pmc_traffic(kernel_substr, args):
    """HBM bytes per launch of the named kernel from the committed PMC passes
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over this same command; counters
    cannot be read from inside the process).  Only valid for the default
    workload the passes were taken on; otherwise None."""
    if (args.workload, args.batch, args.ctx, args.block_size, args.dtype) != ("llama7b", 256, 1024, 16, "float16"):
        return None
    path = os.path.join(ROOT, "profiles", "pmc_hbm_traffic_latest.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    for name, row in table.items():
        if kernel_substr in name:
            return row.get("hbm_traffic_bytes_per_launch")
    return None


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", choices=sorted(WORKLOADS), default="llama7b")
    p.add_argument("--batch", type=int, default=256)
    p.add_argument("--ctx", type=int, default=1024)
    p.add_argument("--block-size", type=int, default=16)
    p.add_argument("--layers", type=int, default=0, help="override the number of layers (0 = model's)")
    p.add_argument("--dtype", choices=["float16", "bfloat16", "float32"], default="float16")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-baseline-seconds", type=float, default=12.0,
                   help="CPU work the cpu_baseline sample is sized to (one layer of the workload, repeated)")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--cpu-plumbing-test", action="store_true",
                   help="run only the multi-process plumbing on CPU (gloo) with a no-op step; used by tests")
    return p.parse_args(argv)


class Dist:
    """One process per GPU; rendezvous from the torch.distributed.run env."""

    def __init__(self, backend: str):
        import torch.distributed as dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = dist
        self.active = self.world > 1
        if self.active:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)

    def barrier(self):
        if self.active:
            self.dist.barrier()

    def max_float(self, x: float, device) -> float:
        if not self.active:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_float(self, x: float, device) -> float:
        if not self.active:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.active:
            self.dist.destroy_process_group()


def timed_region(dist: Dist, device, step_fn, steps: int, warmup: int, sync_fn):
    """W untimed steps, then exactly K steps between barrier+sync fences.
    Returns the MAX elapsed seconds over ranks."""
    for _ in range(warmup):
        step_fn()
    sync_fn()
    dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    return dist.max_float(elapsed, device)


def result_line(args, n_gpus, elapsed, units_all_ranks, unit_name, config, extra):
    ms = elapsed * 1e3 / args.steps
    out = {
        "metric": "output tokens/s through the PagedAttention decode hot path "
                  "(reshape_and_cache + single_query_cached_kv_attention, all layers), "
                  "LLaMA-7B TP=1, batch 256 per GPU",
        "value": units_all_ranks / elapsed,
        "unit": unit_name,
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"float16": "f16", "bfloat16": "bf16", "float32": "f32"}[args.dtype],
        "data": "synthetic",
        "config": config,
    }
    out.update(extra)
    return out


def cpu_plumbing_main(args):
    """No GPU, no kernels: exercises rendezvous, fences, MAX-over-ranks and
    the aggregate arithmetic with a fixed-cost step."""
    import torch
    dist = Dist("gloo")
    device = torch.device("cpu")
    cost = 0.002 * (1 + dist.rank)  # the slowest rank must set the time
    elapsed = timed_region(dist, device, lambda: time.sleep(cost), args.steps, args.warmup, lambda: None)
    units = dist.sum_float(float(args.batch * args.steps), device)
    if dist.rank == 0:
        line = result_line(args, dist.world, elapsed, units, "tokens/s",
                           {"workload": "cpu plumbing test (no-op step)"}, {})
        print(json.dumps(line))
    dist.close()


def main(argv=None):
    args = parse_args(argv)
    if args.cpu_plumbing_test:
        return cpu_plumbing_main(args)

    import numpy as np
    import torch
    import __graft_entry__

    H, D, L_model, desc = WORKLOADS[args.workload]
    L = args.layers or L_model
    S, CTX, B = args.batch, args.ctx, args.block_size
    tdtype = {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32}[args.dtype]
    es = torch.tensor([], dtype=tdtype).element_size()
    x = 16 // es

    dist = Dist("nccl")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(dist.local_rank)
    device = torch.device("cuda", dist.local_rank)
    pkg = __graft_entry__.load_package()
    attention_ops, cache_ops = pkg.attention_ops, pkg.cache_ops

    # ---- synthetic inputs, resident in HBM before timing ----------------------
    torch.manual_seed(0)
    blocks_per_seq = (CTX + B - 1) // B
    NB = S * blocks_per_seq
    cache_bytes = 2 * NB * H * D * B * es
    free_b, _ = torch.cuda.mem_get_info(device)
    if L * cache_bytes > 0.85 * free_b:
        L_fit = max(1, int(0.85 * free_b // cache_bytes))
        if dist.rank == 0:
            print(f"# {L} layers of KV cache ({L * cache_bytes / 2**30:.0f} GiB) do not fit; using {L_fit}",
                  file=sys.stderr)
        L = L_fit
    key_caches, value_caches = [], []
    for _ in range(L):
        kc = torch.empty(NB, H, D // x, B, x, dtype=tdtype, device=device)
        vc = torch.empty(NB, H, D, B, dtype=tdtype, device=device)
        kc.normal_(0.0, 0.5)
        vc.normal_(0.0, 0.5)
        key_caches.append(kc)
        value_caches.append(vc)
    qkv = torch.empty(S, 3, H, D, dtype=tdtype, device=device).normal_(0.0, 0.5)
    query, key, value = qkv.unbind(dim=1)  # strided slices of the fused projection
    out = torch.empty(S, H, D, dtype=tdtype, device=device)
    g = torch.Generator(device="cpu")
    g.manual_seed(0)
    block_tables = torch.randperm(NB, generator=g).to(torch.int32).reshape(S, blocks_per_seq).to(device)
    ctx_list = [CTX] * S
    context_lens = torch.tensor(ctx_list, dtype=torch.int32, device=device)
    pos = CTX - 1  # the new token is the last of the context
    slot_mapping = (block_tables[:, pos // B].to(torch.int64) * B + pos % B).to(torch.int32)
    scale = 1.0 / D ** 0.5

    # The layers of a step share context_lens, hence the launch plan that depends on them
    # alone: layer 0 makes it into the step's workspace, the others reuse it (what
    # PagedAttention.forward of this package does through the step's InputMetadata).
    ws_bytes = attention_ops.decode_workspace_bytes(S, H, D, B, CTX)
    workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=device)

    def step():
        for l in range(L):
            cache_ops.reshape_and_cache(key, value, key_caches[l], value_caches[l], slot_mapping)
            attention_ops.single_query_cached_kv_attention_planned(
                out, query, key_caches[l], value_caches[l], scale, block_tables, context_lens, B, CTX,
                workspace, l > 0, S * CTX)

    elapsed = timed_region(dist, device, step, args.steps, args.warmup, torch.cuda.synchronize)
    units = dist.sum_float(float(S * args.steps), device)

    extra = {}
    # ---- roofline of the dominant kernel (decode attention) --------------------
    if not args.no_roofline and dist.rank == 0:
        n_launch = max(2 * L, 20)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(n_launch)]
        torch.cuda.synchronize()
        for i, (e0, e1) in enumerate(evs):
            l = i % L
            e0.record()
            attention_ops.single_query_cached_kv_attention_planned(
                out, query, key_caches[l], value_caches[l], scale, block_tables, context_lens, B, CTX,
                workspace, True, S * CTX)  # the plan of the timed steps above is still in the workspace
            e1.record()
        torch.cuda.synchronize()
        times_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        mean_ms = sum(times_ms) / len(times_ms)
        nbytes = algorithmic_bytes(ctx_list, H, D, B, es)
        achieved = nbytes / (mean_ms * 1e-3) / 1e9
        # The same kernel inside the step, i.e. right after reshape_and_cache has written this
        # layer's new K/V: those launches run a few percent slower (the first read of the
        # just-written lines), and a kernel trace of this command averages over both kinds.
        evs2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2 * L)]
        for i, (e0, e1) in enumerate(evs2):
            l = i % L
            cache_ops.reshape_and_cache(key, value, key_caches[l], value_caches[l], slot_mapping)
            e0.record()
            attention_ops.single_query_cached_kv_attention_planned(
                out, query, key_caches[l], value_caches[l], scale, block_tables, context_lens, B, CTX,
                workspace, True, S * CTX)
            e1.record()
        torch.cuda.synchronize()
        in_step_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs2) / len(evs2)
        extra["roofline"] = {
            "kernel": "pa_decode_kernel (single_query_cached_kv_attention)",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": pmc_traffic("pa_decode_kernel", args),
            "bytes_per_launch": nbytes,
            "launch_ms_mean": mean_ms,
            "launch_ms_median": times_ms[len(times_ms) // 2],
            "launch_ms_min": times_ms[0],
            "launches_timed": n_launch,
            "in_step_launch_ms_mean": in_step_ms,
            "in_step_achieved": nbytes / (in_step_ms * 1e-3) / 1e9,
        }

    # ---- CPU baseline: the C oracle on this host, one layer of the workload ----
    if not args.no_cpu_baseline and dist.rank == 0 and dist.world == 1:
        from oracle import pa_oracle
        threads = max(1, min(16, os.cpu_count() or 1))
        os.environ["OMP_NUM_THREADS"] = str(threads)
        dname = args.dtype
        def to_np(t):
            t = t.cpu().contiguous()
            return t.view(torch.int16).numpy().view(np.uint16) if dname == "bfloat16" else t.numpy()
        kc_np, vc_np = to_np(key_caches[0]), to_np(value_caches[0])
        q_np = to_np(qkv)[:, 0]
        k_np, v_np = to_np(qkv)[:, 1], to_np(qkv)[:, 2]
        bt_np, cl_np = block_tables.cpu().numpy(), context_lens.cpu().numpy()
        sm_np = slot_mapping.cpu().numpy()
        pa_oracle.lib()
        # One layer's caches (4 GiB per pass: nothing stays in the CPU caches), walked
        # until about 12 s of CPU work are on the clock: one untimed pass sizes the sample.
        def one_layer():
            pa_oracle.reshape_and_cache(k_np, v_np, kc_np, vc_np, sm_np, dname)
            return pa_oracle.single_query_cached_kv_attention(q_np, kc_np, vc_np, scale, bt_np, cl_np, dname)
        t0 = time.perf_counter()
        ref = one_layer()
        t_first = time.perf_counter() - t0
        reps = max(3, min(64, int(args.cpu_baseline_seconds / max(t_first, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(reps):
            ref = one_layer()
        t_layer = (time.perf_counter() - t0) / reps
        extra["cpu_baseline"] = {
            "value": S / (t_layer * L),
            "unit": "tokens/s",
            "cores": threads,
            "kind": "port",
            "sample": f"C oracle (OpenMP, {threads} threads) on one layer of the same workload "
                      f"({S} seqs x {H} heads x ctx {CTX}), {reps} passes, {t_layer:.2f} s per layer; "
                      f"a step is {L} layers",
            "layer_seconds": t_layer,
        }
        # the GPU result of layer 0 doubles as an end-of-bench parity check
        attention_ops.single_query_cached_kv_attention(
            out, query, key_caches[0], value_caches[0], scale, block_tables, context_lens, B, CTX)
        torch.cuda.synchronize()
        got = out.float().cpu().numpy()
        want = (torch.from_numpy(ref.view(np.int16)).view(torch.bfloat16).float().numpy()
                if dname == "bfloat16" else ref.astype(np.float32))
        err = float(np.abs(got - want).max())
        extra["cpu_baseline"]["max_abs_err_gpu_vs_cpu"] = err

    if dist.rank == 0:
        config = {
            "workload": f"{desc}: {L} layers x (reshape_and_cache + single_query_cached_kv_attention), "
                        f"batch {S} per GPU, {H} heads x {D}, block {B}, ctx {CTX}, "
                        f"distinct physical blocks, per-layer 2x{cache_bytes / 2 / 2**30:.2f} GiB caches",
            "batch_per_gpu": S, "num_heads": H, "head_size": D, "block_size": B, "context_len": CTX,
            "layers": L, "sharding": "independent replicas per GPU (no data-path collective)",
        }
        print(json.dumps(result_line(args, dist.world, elapsed, units, "tokens/s", config, extra)))
    dist.close()


if __name__ == "__main__":
    main()
