#!/usr/bin/env python3
"""Benchmark of the PagedAttention decode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--tp]

Metric (BASELINE.json): output tokens/s + paged-attention HBM GB/s, LLaMA-7B
TP=1, batch 256.  One STEP is one decode iteration of the hot path over the
whole model: for each of the 32 layers, `cache_ops.reshape_and_cache` writes
the batch's new K/V vectors into that layer's paged cache and
`attention_ops.single_query_cached_kv_attention_planned` attends over it
(batch 256, 32 heads x 128, block 16, context 1024, fp16; every logical block a
distinct physical block; one 2 GiB K and one 2 GiB V cache per layer, all
resident in HBM before timing starts).  The attention's launch plan (the order
in which rows are dealt to the XCDs) depends on the context lengths only and is
made once per step, by layer 0, as in this package's PagedAttention layer.
value = tokens produced per second by that path = batch * steps / elapsed.  The
GEMMs, sampler and scheduler of the full model are outside this path.

N > 1 (launched by torch.distributed.run, one rank per GPU), two forms:

  default   every rank runs the whole per-GPU workload on its own caches: the
            (sequence, head) units are independent, no data-path collective;
            value aggregates all ranks ("scaling": "weak").  The line also
            carries a "tp" object: the head-sharded form below measured in the
            same process with the backend's all-reduce, and "tp_push": the same
            step with this package's own all-reduce, measured by one child
            process per rank (a fault there cannot take the line with it).
  --tp      the reference's tensor parallelism: rank r owns heads
            [r*H/N, (r+1)*H/N) and their K/V; per layer reshape_and_cache +
            attention on the local heads, the row-parallel o_proj matmul into
            the all-reduce launcher's buffer, and the RCCL all-reduce of
            [batch, hidden] on a side HIP stream, overlapped with the next
            layer's cache write and attention.  Total work is fixed
            ("scaling": "strong"); value = batch * steps / elapsed.

Timing is barrier + synchronize on both sides, MAX over ranks.

The JSON line also carries
  roofline      decode kernel: algorithmic bytes per launch / MEDIAN launch
                time measured with HIP events on the launch stream, against the
                8 TB/s HBM3E peak; the same for the operator with the
                reference's signature; `traffic` = PMC bytes of a committed
                rocprofv3 pass over this command (`traffic_source` names it);
  c2            the same measurement on the microbenchmark shape of
                BASELINE.json configs[1] (12 heads x 64);
  cpu_baseline  CPU statements of the path timed on this host, rank 0, N=1: the
                C oracle on one layer of the workload, and two torch-CPU
                statements (per-token loop, vectorised gather) on a slice;
  and, at N = 1 on the default workload, four secondary measurements of the
  operators either side of the decode kernel, each guarded so that a failure
  becomes an "error" entry instead of costing the line:
  cache_write   reshape_and_cache at the step's size: us per launch (HIP-graph
                replay over the layers), its algorithmic bytes, and the PMC
                traffic of the committed rocprofv3 pass next to them;
  small_batch   batches of 1 / 4 / 16 sequences through the operator with the
                reference's signature (graph replay): us per launch, fraction
                of the HBM peak;
  prompt        prompt attention (the operator behind the xformers call of
                the reference) on four LLaMA-7B prompt batches, bf16: us,
                TFLOP/s, fraction of the 2.5 PFLOP/s dense MFMA peak;
  decoder_step  a whole LLaMA-7B decode step on the same caches — torch GEMMs,
                embedding, argmax, and this package's operators for the rest:
                ms per step and tokens/s, eager and from a HIP graph.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak
MFMA_PEAK_TFLOPS_16BIT = 2500.0  # dense fp16 / bf16 matrix peak (without the 2:1 sparsity of the headline figure)

WORKLOADS = {
    # name: (num_heads, head_size, num_layers, description)
    "llama7b": (32, 128, 32, "LLaMA-7B"),
    "llama13b": (40, 128, 40, "LLaMA-13B"),
    "neox20b": (64, 96, 44, "GPT-NeoX-20B"),
    "opt125m_c2": (12, 64, 12, "OPT-125m (microbench shape of BASELINE.json configs[1])"),
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


PMC_TRAFFIC_FILE = os.path.join("profiles", "pmc_hbm_traffic_latest.json")


def pmc_traffic(kernel_substr, args):
    """(HBM bytes per launch of the named kernel, where that figure comes from).
    Hardware counters cannot be read from inside the process: the figure is from
    the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) over this same
    command, valid only for the default workload; otherwise (None, None)."""
    if (args.workload, args.batch, args.ctx, args.block_size, args.dtype) != ("llama7b", 256, 1024, 16, "float16"):
        return None, None
    try:
        with open(os.path.join(ROOT, PMC_TRAFFIC_FILE)) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None, None
    fresh = pmc_sources_unchanged(table)
    note = {True: "; the kernels' sources are unchanged since that pass",
            False: "; STALE: the kernels' sources have changed since that pass",
            None: ""}[fresh]
    for name, row in table.items():
        if not name.startswith("_") and kernel_substr in name:
            return row.get("hbm_traffic_bytes_per_launch"), \
                f"{PMC_TRAFFIC_FILE} (committed rocprofv3 --pmc pass over this command, not this run{note})"
    return None, None


def pmc_sources_unchanged(table):
    """Whether the sources of the kernels counted in the committed PMC file are the ones in this tree
    (the file records their hashes); None if it holds no hashes or a source cannot be read."""
    import hashlib
    recorded = table.get("_kernel_sources_sha16")
    if not isinstance(recorded, dict) or not recorded:
        return None
    try:
        return all(hashlib.sha256(open(os.path.join(ROOT, "vllm-v0.1.0_amd", "csrc", name), "rb").read()).hexdigest()[:16] == sha
                   for name, sha in recorded.items())
    except OSError:
        return None


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--tp", action="store_true",
                   help="N > 1: shard heads over the ranks (tensor parallel) with the o_proj all-reduce in the step")
    p.add_argument("--tp-emulate", type=int, default=0, metavar="N",
                   help="one GPU: run rank 0's share of a tensor-parallel group of N (heads/N, o_proj shard) with the "
                        "collective replaced by a local add of a same-sized 'remote' partial sum: the shard's device "
                        "time without xGMI; the line says so")
    p.add_argument("--workload", choices=sorted(WORKLOADS), default="llama7b")
    p.add_argument("--batch", type=int, default=256)
    p.add_argument("--ctx", type=int, default=1024)
    p.add_argument("--block-size", type=int, default=16)
    p.add_argument("--layers", type=int, default=0, help="override the number of layers (0 = model's)")
    p.add_argument("--dtype", choices=["float16", "bfloat16", "float32"], default="float16")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-baseline-seconds", type=float, default=10.0,
                   help="CPU work the C-oracle sample is sized to (one layer of the workload, repeated)")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-c2", action="store_true", help="skip the secondary microbench-shape measurement")
    p.add_argument("--no-secondary", action="store_true",
                   help="skip cache_write / small_batch / prompt / decoder_step (N = 1, default workload only)")
    p.add_argument("--no-tp-leg", action="store_true", help="N > 1 without --tp: skip the extra head-sharded measurement")
    p.add_argument("--tp-leg-deadline", type=float, default=180.0,
                   help="N > 1 without --tp: seconds the extra head-sharded legs may take before the line is printed without them")
    p.add_argument("--tp-graph", action="store_true",
                   help="tensor-parallel legs: replay the all-reduce from a HIP graph captured on the side stream "
                        "(default: direct RCCL calls on the side stream; the graph path has only been exercised with a "
                        "stand-in collective on one GPU)")
    p.add_argument("--tp-collective", choices=["backend", "push"], default="backend",
                   help="tensor-parallel legs: 'backend' = all_reduce of --dist-backend (RCCL); 'push' = this package's "
                        "one-shot all-reduce over peer-mapped inboxes (worker/push_all_reduce.py), which also runs with "
                        "several ranks on one GPU")
    p.add_argument("--rehearse-child-abort", type=int, default=-1, metavar="RANK",
                   help="rehearsal: the push leg's child process of this rank aborts before it times anything (the "
                        "line must still come out, with the leg marked failed)")
    p.add_argument("--abort-before-timing", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--roofline-only", action="store_true",
                   help="launch nothing but isolated decode kernels (5 warm-ups + 40 timed): the command to put "
                        "under rocprofv3 --kernel-trace so that its per-dispatch median reproduces roofline.frac")
    p.add_argument("--progress", action="store_true", help="print a line per phase and rank on stderr")
    p.add_argument("--dist-backend", default="nccl",
                   help="torch.distributed backend for N > 1 (nccl = RCCL).  gloo lets several ranks share one GPU: a "
                        "rehearsal of the multi-rank control flow on a one-GPU box, not a measurement")
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

    def describe(self, torch):
        """Who took part, for a reader of the line: the backend, the ranks, the GPUs they sat on (gathered
        over the group, so a rank that saw another device shows) and the collective library's version."""
        info = {"backend": self.dist.get_backend() if self.active else None, "world": self.world}
        mine = {"rank": self.rank, "device": torch.cuda.current_device() if torch.cuda.is_available() else None,
                "visible": torch.cuda.device_count() if torch.cuda.is_available() else 0}
        seen = [None] * self.world
        if self.active:
            self.dist.all_gather_object(seen, mine)
        else:
            seen = [mine]
        info["devices_seen"] = seen
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            info["rccl_version"] = f"unavailable ({type(e).__name__})"
        return info

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


def metric_text(args, model_name, tp):
    form = f"TP={tp}" if tp > 1 else "TP=1"
    per = "" if tp > 1 else " per GPU"
    return ("output tokens/s through the PagedAttention decode hot path "
            "(reshape_and_cache + single_query_cached_kv_attention, all layers), "
            f"{model_name} {form}, batch {args.batch}{per}")


def result_line(args, n_gpus, elapsed, units, unit_name, config, extra, metric, scaling="weak"):
    ms = elapsed * 1e3 / args.steps
    out = {
        "metric": metric,
        "value": units / elapsed,
        "unit": unit_name,
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms,
        "higher_is_better": True,
        "scaling": scaling,
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
    if args.tp:  # strong scaling: every rank works on the same tokens
        units, scaling = float(args.batch * args.steps), "strong"
    else:
        units, scaling = dist.sum_float(float(args.batch * args.steps), device), "weak"
    if dist.rank == 0:
        line = result_line(args, dist.world, elapsed, units, "tokens/s",
                           {"workload": "cpu plumbing test (no-op step)"}, {},
                           metric_text(args, "no model", dist.world if args.tp else 1), scaling)
        print(json.dumps(line))
    dist.close()


# ---- measurement helpers -----------------------------------------------------------

def time_launches(torch, fn, n, warmup=3):
    """Sorted per-launch milliseconds of n calls of fn(i), HIP events on the current stream."""
    for i in range(warmup):
        fn(i)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for i, (e0, e1) in enumerate(evs):
        e0.record()
        fn(i)
        e1.record()
    torch.cuda.synchronize()
    return sorted(e0.elapsed_time(e1) for e0, e1 in evs)


def median(xs):
    return xs[len(xs) // 2]


class Workload:
    """Synthetic decode-step inputs of one (sharded) model, resident in HBM."""

    def __init__(self, torch, device, H, D, L, S, CTX, B, tdtype, free_fraction=0.85, rank=0, note=None):
        es = torch.tensor([], dtype=tdtype).element_size()
        x = 16 // es
        self.H, self.D, self.S, self.CTX, self.B, self.es = H, D, S, CTX, B, es
        blocks_per_seq = (CTX + B - 1) // B
        NB = S * blocks_per_seq
        self.cache_bytes = 2 * NB * H * D * B * es
        free_b, _ = torch.cuda.mem_get_info(device)
        if L * self.cache_bytes > free_fraction * free_b:
            L_fit = max(1, int(free_fraction * free_b // self.cache_bytes))
            if rank == 0:
                print(f"# {L} layers of KV cache ({L * self.cache_bytes / 2**30:.0f} GiB) do not fit; using {L_fit}",
                      file=sys.stderr)
            L = L_fit
        self.L = L
        # placed the way this package's CacheEngine places a layer's caches (the value cache
        # KV_STAGGER_BYTES behind the end of the key cache; worker/cache_engine.py says why)
        from vllm_amd.worker import carve_kv_cache, kv_slab_elements
        self.key_caches, self.value_caches = [], []
        for _ in range(L):
            slab = torch.empty(kv_slab_elements(1, NB, (H, D, B), es), dtype=tdtype, device=device)
            slab.normal_(0.0, 0.5)
            kc, vc = carve_kv_cache(slab, 0, NB, (H, D // x, B, x), (H, D, B))
            self.key_caches.append(kc)
            self.value_caches.append(vc)
        self.qkv = torch.empty(S, 3, H, D, dtype=tdtype, device=device).normal_(0.0, 0.5)
        self.query, self.key, self.value = self.qkv.unbind(dim=1)  # strided slices of the fused projection
        self.out = torch.empty(S, H, D, dtype=tdtype, device=device)
        g = torch.Generator(device="cpu")
        g.manual_seed(0)  # the same tables on every rank, as the engine hands every worker the same scheduler output
        self.block_tables = torch.randperm(NB, generator=g).to(torch.int32).reshape(S, blocks_per_seq).to(device)
        self.ctx_list = [CTX] * S
        self.context_lens = torch.tensor(self.ctx_list, dtype=torch.int32, device=device)
        pos = CTX - 1  # the new token is the last of the context
        self.slot_mapping = (self.block_tables[:, pos // B].to(torch.int64) * B + pos % B).to(torch.int32)
        self.scale = 1.0 / D ** 0.5
        self.total_ctx = S * CTX

    def bytes_per_launch(self):
        return algorithmic_bytes(self.ctx_list, self.H, self.D, self.B, self.es)


def decode_roofline(torch, attention_ops, cache_ops, w, workspace, n_launch):
    """Isolated launches of the decode operator on workload w, timed one by one."""
    def planned(i):
        l = i % w.L
        attention_ops.single_query_cached_kv_attention_planned(
            w.out, w.query, w.key_caches[l], w.value_caches[l], w.scale, w.block_tables, w.context_lens, w.B, w.CTX,
            workspace, True, w.total_ctx)

    # The reference's own CacheEngine makes a torch.empty for each cache (vllm/worker/cache_engine.py:65-79):
    # the reference-signature launches run on caches allocated that way when memory allows (three
    # layers' worth, walked in turn), not on this package's staggered slab.
    ref_caches = None
    try:
        free_b, _ = torch.cuda.mem_get_info(w.query.device)
        if 3 * w.cache_bytes < 0.5 * free_b:
            ref_caches = [(torch.empty_like(w.key_caches[0]).normal_(0.0, 0.5), torch.empty_like(w.value_caches[0]).normal_(0.0, 0.5))
                          for _ in range(3)]
    except Exception:  # noqa: BLE001
        ref_caches = None

    def reference_signature(i):
        kc, vc = ref_caches[i % 3] if ref_caches is not None else (w.key_caches[i % w.L], w.value_caches[i % w.L])
        attention_ops.single_query_cached_kv_attention(
            w.out, w.query, kc, vc, w.scale, w.block_tables, w.context_lens, w.B, w.CTX)

    # make the plan the timed launches reuse
    attention_ops.single_query_cached_kv_attention_planned(
        w.out, w.query, w.key_caches[0], w.value_caches[0], w.scale, w.block_tables, w.context_lens, w.B, w.CTX,
        workspace, False, w.total_ctx)
    times = time_launches(torch, planned, n_launch)
    ref_times = time_launches(torch, reference_signature, n_launch)
    ref_alloc = "two torch.empty per layer, as the reference's CacheEngine allocates" if ref_caches is not None else \
        "the workload's caches (no room for separately allocated ones)"
    ref_caches = None
    nbytes = w.bytes_per_launch()
    med = median(times)
    achieved = nbytes / (med * 1e-3) / 1e9
    return {
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS,
        "bytes_per_launch": nbytes,
        "launch_ms_median": med,
        "launch_ms_mean": sum(times) / len(times),
        "launch_ms_min": times[0],
        "launch_ms_max": times[-1],
        "launches_timed": n_launch,
        "timing": "HIP events around each launch on the launch stream; achieved / frac use the median",
        "reference_signature_launch_ms": median(ref_times),
        "reference_signature_achieved": nbytes / (median(ref_times) * 1e-3) / 1e9,
        "reference_signature_note": "attention_ops.single_query_cached_kv_attention(out, query, key_cache, value_cache, "
                                    "scale, block_tables, context_lens, block_size, max_context_len): what the unchanged "
                                    "reference caller invokes; plans per call, scratch from torch's allocator; caches: "
                                    + ref_alloc,
    }


def graph_time_us(torch, fn, per_graph, replays=7):
    """Median microseconds per call of fn(i), i = 0 .. per_graph - 1, replayed from one HIP graph
    (the host's launch cost is not in the figure)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(min(per_graph, 3)):
            fn(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(per_graph):
            fn(i)
    graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(replays):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graph.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / per_graph * 1e3)
    return median(sorted(ts))


def guarded(extra, key, fn):
    """extra[key] = fn(), or what went wrong: a secondary measurement never costs the line."""
    try:
        extra[key] = fn()
    except Exception as e:  # noqa: BLE001
        extra[key] = {"error": f"{type(e).__name__}: {e}"[:400]}


def cache_write_measurement(torch, cache_ops, args, w):
    """reshape_and_cache of the step's new tokens, one launch per layer, replayed from a graph."""
    us = graph_time_us(torch, lambda i: cache_ops.reshape_and_cache(
        w.key, w.value, w.key_caches[i % w.L], w.value_caches[i % w.L], w.slot_mapping), w.L)
    moved = 2 * 2 * w.S * w.H * w.D * w.es + 4 * w.S  # K and V rows read, the same bytes written, the slots read
    traffic, source = pmc_traffic("pa_token_append_kernel", args)
    # the kernel rewrites the whole value tile of every block it appends to (full lines instead of one
    # 2-byte element in each of the tile's rows): what it has to move is the K rows in and out, the V
    # rows in, and every touched value tile in and out
    tile_bytes = w.H * w.D * w.B * w.es
    moves = 2 * w.S * w.H * w.D * w.es + w.S * w.H * w.D * w.es + 2 * w.S * tile_bytes + 4 * w.S
    res = {"kernel": "pa_token_append_kernel (cache_ops.reshape_and_cache, decode-sized batch)",
           "us_per_launch": us, "tokens": w.S, "algorithmic_bytes": moved,
           "bytes_the_kernel_moves": moves, "achieved_gbps_on_moved_bytes": moves / us / 1e3,
           "frac_of_hbm_peak_on_moved_bytes": moves / us / 1e3 / HBM_PEAK_GBPS,
           "timing": f"HIP-graph replay of {w.L} launches, one per layer's caches; median of 7 replays",
           "traffic": traffic, "traffic_source": source}
    if traffic:
        res["traffic_over_algorithmic"] = traffic / moved
    return res


def small_batch_measurement(torch, attention_ops, w):
    """Few sequences through attention_ops.single_query_cached_kv_attention (the reference's signature:
    sizes its scratch and plans its launch per call), on slices of the workload's caches, a different
    layer's memory for every launch of a replay."""
    res = {"entry": "attention_ops.single_query_cached_kv_attention (reference signature), HIP-graph replay of one "
                    f"launch per layer's caches, {w.H} heads x {w.D}, ctx {w.CTX}, block {w.B}",
           "shapes": []}
    for S in (1, 4, 16):
        if S > w.S:
            continue
        q, out = w.query[:S], w.out[:S]
        bt, cl = w.block_tables[:S].contiguous(), w.context_lens[:S].contiguous()
        us = graph_time_us(torch, lambda i: attention_ops.single_query_cached_kv_attention(
            out, q, w.key_caches[i % w.L], w.value_caches[i % w.L], w.scale, bt, cl, w.B, w.CTX), w.L)
        nbytes = algorithmic_bytes(w.ctx_list[:S], w.H, w.D, w.B, w.es)
        res["shapes"].append({"batch": S, "us_per_launch": us, "bytes_per_launch": nbytes,
                              "achieved_gbps": nbytes / us / 1e3, "frac": nbytes / us / 1e3 / HBM_PEAK_GBPS})
    return res


def prompt_measurement(torch, attention_ops, device, H, D):
    """Prompt attention on LLaMA-7B-shaped prompt batches (bf16): causal FLOPs = 4 H D sum n (n + 1) / 2."""
    res = {"kernel": "pa_prompt_kernel (attention_ops.multi_query_kv_attention)", "dtype": "bf16", "bound": "mfma",
           "peak": MFMA_PEAK_TFLOPS_16BIT, "unit": "TFLOP/s", "heads": H, "head_size": D,
           "timing": "HIP-graph replay of 5 launches; median of 7 replays", "shapes": []}
    for lens in ([512] * 5, [2560], [2048] * 8, [16384]):
        T = sum(lens)
        qkv = torch.randn(T, 3, H, D, dtype=torch.bfloat16, device=device)
        q, k, v = qkv.unbind(dim=1)
        out = torch.empty(T, H, D, dtype=torch.bfloat16, device=device)
        us = graph_time_us(torch, lambda i: attention_ops.multi_query_kv_attention(out, q, k, v, lens, D ** -0.5), 5)
        flops = 4 * H * D * sum(n * (n + 1) // 2 for n in lens)
        # q, k, v read once and out written once: at short prompts that, not the MFMA pipe, is the lower bound
        nbytes = 4 * T * H * D * 2
        floor_us = max(flops / MFMA_PEAK_TFLOPS_16BIT / 1e6, nbytes / HBM_PEAK_GBPS / 1e3)
        res["shapes"].append({"prompts": f"{len(lens)} x {lens[0]}", "tokens": T, "us_per_launch": us, "flops": flops,
                              "achieved": flops / us / 1e6, "frac": flops / us / 1e6 / MFMA_PEAK_TFLOPS_16BIT,
                              "hbm_bytes": nbytes, "floor_us": floor_us, "frac_of_floor": floor_us / us})
        del qkv, out
    return res


def decoder_step_measurement(pkg, w):
    """A whole LLaMA-7B decode step around the path, on the workload's caches and block tables."""
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import bench_decoder_step
    r = bench_decoder_step.decoder_step(pkg, "7b", batch=w.S, ctx=w.CTX, steps=5,
                                        caches=list(zip(w.key_caches, w.value_caches)), block_tables=w.block_tables)
    r["what"] = ("embedding, 32 x (rms_norm, qkv GEMM, rotary + reshape_and_cache, single_query_cached_kv_attention, o_proj "
                 "GEMM, rms_norm, gate_up GEMM, silu_and_mul, down GEMM), final norm, logits GEMM, argmax; random weights; "
                 "torch GEMMs, this package's operators for the rest")
    return r


def tp_leg(torch, dist, device, pkg, args, H, D, L, tdtype, model_name, shared_caches=None, emulate=0, collective=None):
    """The head-sharded step: per layer reshape_and_cache + attention on this rank's heads, the
    row-parallel o_proj matmul into the launcher's ring buffer, the all-reduce on the side stream.
    Returns (elapsed seconds MAX over ranks, info dict)."""
    from vllm_amd.worker import GraphAllReduce, TensorParallelDecodeStep, head_shard
    tp, rank = (emulate, 0) if emulate else (dist.world, dist.rank)
    lo, hi = head_shard(rank, tp, H)
    Hl = hi - lo
    hidden = H * D
    S, CTX, B = args.batch, args.ctx, args.block_size
    if shared_caches is None:
        w = Workload(torch, device, Hl, D, L, S, CTX, B, tdtype, rank=rank)
    else:
        # carve the local-heads caches out of the replica leg's allocations: no second set of caches
        w = shared_caches
        x = 16 // w.es
        NB = w.key_caches[0].shape[0]
        n = NB * Hl * D * B
        kcs = [kc.view(-1)[:n].view(NB, Hl, D // x, B, x) for kc in w.key_caches]
        vcs = [vc.view(-1)[:n].view(NB, Hl, D, B) for vc in w.value_caches]
        local = Workload.__new__(Workload)
        local.__dict__.update(w.__dict__)
        local.H, local.key_caches, local.value_caches = Hl, kcs, vcs
        local.qkv = w.qkv[:, :, :Hl].contiguous()  # the local fused projection holds the local heads only
        local.query, local.key, local.value = local.qkv.unbind(dim=1)
        local.out = torch.empty(S, Hl, D, dtype=tdtype, device=device)
        w = local
    L = w.L
    w_o = [torch.empty(Hl * D, hidden, dtype=tdtype, device=device).normal_(0.0, hidden ** -0.5) for _ in range(L)]
    reduce_fn = None
    if emulate:
        remote = torch.zeros(S, hidden, dtype=tdtype, device=device)
        reduce_fn = lambda t: t.add_(remote[:t.shape[0]])  # noqa: E731  (stand-in: same bytes, no xGMI)
    push = None
    if not emulate and (collective or args.tp_collective) == "push" and tp > 1:
        from vllm_amd.worker import PushAllReduce
        # handles exchanged over the process group; set-up ends with the collective self-test (both forms, a
        # small and a multi-MiB message): all ranks agree that the peers' memory is reachable and ordered, or
        # all raise.  A bench leg can afford a shorter wait for a dead peer than a serving process.
        push = PushAllReduce(S * hidden * w.es, device=device, wait_seconds=10.0)
        reduce_fn = push
    elif not emulate and args.dist_backend != "nccl":
        # rehearsal on a box without one GPU per rank: the sum goes through host memory (the
        # backend's own reduction of device tensors is not what is being rehearsed)
        def reduce_fn(t):  # noqa: F811
            host = t.cpu()
            dist.dist.all_reduce(host)
            t.copy_(host)
    use_graph = bool(args.tp_graph or emulate)
    launcher = GraphAllReduce(S, hidden, tdtype, disable_graph=not use_graph, device=device, reduce_fn=reduce_fn)
    launcher.precapture([S])  # no capture inside the timed loop (a no-op without graphs)
    step_obj = TensorParallelDecodeStep(rank, tp, H, D, w.scale, launcher, pkg.attention_ops, pkg.cache_ops)
    dense_out = torch.empty(S, hidden, dtype=tdtype, device=device) if tp == 1 else None

    def step():
        step_obj.begin_step()
        pend = []
        for l in range(L):
            p = step_obj.layer(w.query, w.key, w.value, w.key_caches[l], w.value_caches[l], w.slot_mapping,
                               w.block_tables, w.context_lens, CTX, w_o[l], w.out, w.total_ctx)
            if p is None:  # one rank: the same matmul, nothing to sum
                torch.matmul(w.out.view(S, Hl * D), w_o[l], out=dense_out)
            else:
                pend.append(p)
        for p in pend[-2:]:  # earlier ones were fenced when their ring buffer came round again
            p.wait()

    progress(dist, f"tp leg: launcher ready ({'graph' if use_graph else 'direct'}); timing")
    if args.abort_before_timing:
        os.abort()  # rehearsal of a rank that dies in the leg
    elapsed = timed_region(dist, device, step, args.steps, args.warmup, torch.cuda.synchronize)
    # a reduce that gave up on a peer left NaNs: no number from such a run, and all ranks leave the leg together
    failure = None
    try:
        launcher.check()
    except RuntimeError as e:
        failure = str(e)
    if (dist.sum_float(1.0 if failure else 0.0, device) if not emulate else bool(failure)):
        raise RuntimeError(failure or "another rank's all-reduce gave up waiting for a peer")
    progress(dist, f"tp leg timed: {elapsed * 1e3 / args.steps:.3f} ms")
    coll = ("one-shot push" if push is not None else "stand-in" if emulate else
            "RCCL" if args.dist_backend == "nccl" else f"{args.dist_backend} (host-staged)")
    info = {
        "tp": tp, "heads_per_rank": Hl, "layers": L, "scaling": "strong", **({} if emulate else dist.describe(torch)),
        "value": S * args.steps / elapsed, "unit": "tokens/s", "ms_per_step": elapsed * 1e3 / args.steps,
        "step": "per layer: reshape_and_cache + single_query_cached_kv_attention on the local heads, row-parallel "
                "o_proj matmul [batch, heads/tp * head_size] x [.., hidden] into the all-reduce buffer, "
                f"{coll} all-reduce of [batch, hidden] on a side HIP stream (event hand-off"
                f"{', HIP graph replay' if use_graph else ''}), overlapped with the next layer's cache write and attention",
        "all_reduce_bytes": S * hidden * w.es, "all_reduces_per_step": L if tp > 1 else 0,
        "all_reduce_launch": "HIP graph replay on the side stream" if use_graph else "direct call on the side stream",
        "all_reduce_collective": ("one-shot push over peer-mapped inboxes (pa_comm_all_reduce)" if push is not None else
                                  "local stand-in" if emulate else f"{args.dist_backend} all_reduce"
                                  + ("" if args.dist_backend == "nccl" else " staged through host memory")),
    }
    if emulate:
        info["emulated"] = ("one GPU ran rank 0's share; the all-reduce is replaced by a local add of a same-sized "
                            "buffer on the side stream: device time of the shard, no xGMI traffic")
    if tp > 1:
        # the collective alone: launch + wait, back to back
        def one(i):
            launcher.launch_async(launcher.next_buffer(S)).wait()
        ts = time_launches(torch, one, 40, warmup=5)
        progress(dist, "tp leg: isolated all-reduce timed")
        info["all_reduce_isolated_us_median"] = median(ts) * 1e3
        if not emulate:
            info["all_reduce_isolated_us_max_over_ranks"] = dist.max_float(median(ts) * 1e3, device)
    if push is not None:
        # both forms of the collective alone, for the message of this step: the data PA_COMM_AUTO's threshold needs
        for name, algo in (("one_shot", push.ONE_SHOT), ("two_shot", push.TWO_SHOT)):
            push.algorithm = algo
            if use_graph:
                launcher.graphs.clear()  # captured with the other form
                launcher.precapture([S])
            ts = time_launches(torch, one, 40, warmup=5)
            info[f"all_reduce_isolated_us_{name}"] = dist.max_float(median(ts) * 1e3, device)
        push.algorithm = push.AUTO
        done, gave_up = push.status()
        info["push_reduces_done"], info["push_waits_given_up"] = done, gave_up
        dist.barrier()
        push.close()
    return elapsed, info, w


CHILD = None  # the push leg's child process of this rank, while it runs (the deadline ends it)


def push_leg_in_children(args, dist, device, budget_s):
    """The head-sharded step with this package's own all-reduce, measured by CHILD processes: one
    per rank, with a process group of their own, started as `bench.py --tp --tp-collective push`.
    A multi-GPU box is the first place where those kernels store into another GPU's memory over
    xGMI; a fault there ends the process it happens in, and that must not be the process that
    holds the result line.  Returns the leg's info dict (rank 0) or {"error": ...}."""
    global CHILD
    import socket
    import subprocess
    port = 0.0
    if dist.rank == 0:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = float(s.getsockname()[1])
    port = int(dist.max_float(port, device))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
        del env[k]  # with the agent's store flag set, rank 0 would look for a store nobody serves on the new port
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(dist.world), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--tp", "--tp-collective", "push", "--workload", args.workload,
           "--batch", str(args.batch), "--ctx", str(args.ctx), "--block-size", str(args.block_size),
           "--layers", str(args.layers), "--dtype", args.dtype, "--dist-backend", args.dist_backend,
           "--no-cpu-baseline", "--no-roofline", "--no-c2"]
    cmd += ["--tp-graph"] if args.tp_graph else []
    cmd += ["--progress"] if args.progress else []
    cmd += ["--abort-before-timing"] if args.rehearse_child_abort == dist.rank else []
    CHILD = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        out, err = CHILD.communicate(timeout=max(budget_s, 20.0))
        rc = CHILD.returncode
    except subprocess.TimeoutExpired:
        CHILD.kill()
        out, err = CHILD.communicate()
        rc = "timeout"
    CHILD = None
    if PROGRESS and err:
        sys.stderr.write(err)
    failed = dist.sum_float(0.0 if rc == 0 else 1.0, device)  # also: every rank's child has ended
    where = "measured in child processes (one per rank, a process group of their own)"
    if dist.rank != 0:
        return None
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if rc != 0 or failed or len(lines) != 1:
        return {"error": f"child processes of {int(failed)} of {dist.world} ranks failed; rank 0's child: exit {rc}, "
                         f"stderr tail: {err[-300:]!r}", "where": where}
    line = json.loads(lines[0])
    info = dict(line["tp"], value=line["value"], unit=line["unit"], ms_per_step=line["ms_per_step"], where=where)
    return info


def cpu_baseline(torch, np, args, w, L, attention_ops, cache_ops):
    """CPU statements of the path on this host: C oracle on one layer; two torch statements on a slice."""
    from oracle import pa_oracle, torch_cpu_baseline
    host_cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = host_cores
    # the box gives a one-GPU job a share of the host: OMP_NUM_THREADS is that share where it is set
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = max(1, min(usable, share) if share > 0 else usable)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    torch.set_num_threads(threads)
    dname = args.dtype
    S, H, CTX = w.S, w.H, w.CTX

    def to_np(t):
        t = t.cpu().contiguous()
        return t.view(torch.int16).numpy().view(np.uint16) if dname == "bfloat16" else t.numpy()

    kc_t, vc_t = w.key_caches[0].cpu(), w.value_caches[0].cpu()
    kc_np, vc_np = to_np(kc_t), to_np(vc_t)
    qkv_np = to_np(w.qkv)
    q_np, k_np, v_np = qkv_np[:, 0], qkv_np[:, 1], qkv_np[:, 2]
    bt_np, cl_np = w.block_tables.cpu().numpy(), w.context_lens.cpu().numpy()
    sm_np = w.slot_mapping.cpu().numpy()
    pa_oracle.lib()

    # One layer's caches (4 GiB per pass: nothing stays in the CPU caches), walked until about
    # cpu_baseline_seconds of CPU work are on the clock: one untimed pass sizes the sample.
    def one_layer():
        pa_oracle.reshape_and_cache(k_np, v_np, kc_np, vc_np, sm_np, dname)
        return pa_oracle.single_query_cached_kv_attention(q_np, kc_np, vc_np, w.scale, bt_np, cl_np, dname)
    t0 = time.perf_counter()
    ref = one_layer()
    t_first = time.perf_counter() - t0
    reps = max(3, min(64, int(args.cpu_baseline_seconds / max(t_first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        ref = one_layer()
    t_layer = (time.perf_counter() - t0) / reps
    nbytes = w.bytes_per_launch()
    res = {
        "value": S / (t_layer * L),
        "unit": "tokens/s",
        "cores": threads,
        "host_cores": host_cores,
        "cores_note": f"{threads} threads = this job's share of the host (OMP_NUM_THREADS / affinity); "
                      f"os.cpu_count() of the machine = {host_cores}",
        "kind": "port",
        "sample": f"C oracle (OpenMP, {threads} threads) on one layer of the same workload "
                  f"({S} seqs x {H} heads x ctx {CTX}), {reps} passes, {t_layer:.2f} s per layer; "
                  f"a step is {L} layers",
        "layer_seconds": t_layer,
        "gbps": nbytes / t_layer / 1e9,
    }
    # the GPU result of layer 0 doubles as an end-of-bench parity check: the same two operators as one_layer(), on
    # the same cache contents (the secondary measurements before this ran a decoder step of their own on these
    # caches, which left its keys and values in the new tokens' slots)
    cache_ops.reshape_and_cache(w.key, w.value, w.key_caches[0], w.value_caches[0], w.slot_mapping)
    attention_ops.single_query_cached_kv_attention(
        w.out, w.query, w.key_caches[0], w.value_caches[0], w.scale, w.block_tables, w.context_lens, w.B, CTX)
    torch.cuda.synchronize()
    got = w.out.float().cpu().numpy()
    want = (torch.from_numpy(ref.view(np.int16)).view(torch.bfloat16).float().numpy()
            if dname == "bfloat16" else ref.astype(np.float32))
    res["max_abs_err_gpu_vs_cpu"] = float(np.abs(got - want).max())
    tol = 1e-2 if dname == "bfloat16" else 1e-3  # outputs are means of ~ctx values of N(0, 0.5): |out| ~ 0.02-0.1
    res["parity_tolerance"] = tol
    res["parity_ok"] = bool(res["max_abs_err_gpu_vs_cpu"] <= tol)

    # torch on the CPU, fp32 accumulation, the decode attention of a slice of the same layer
    # (the cache write is not part of these statements)
    q_t = w.qkv.cpu()[:, 0]
    bt_t, cl_t = w.block_tables.cpu(), w.context_lens.cpu()
    for name, fn, n_seqs in (("torch_loop", torch_cpu_baseline.decode_loop, 4),
                             ("torch_vectorised", torch_cpu_baseline.decode_vectorised, 32)):
        n_seqs = min(n_seqs, S)
        t0 = time.perf_counter()
        o = fn(q_t[:n_seqs], kc_t, vc_t, bt_t[:n_seqs], cl_t[:n_seqs], w.scale)
        dt = time.perf_counter() - t0
        b = algorithmic_bytes(w.ctx_list[:n_seqs], H, w.D, w.B, w.es)
        res[name] = {
            "sample": f"decode attention of {n_seqs} of the {S} sequences of one layer, {threads} torch threads, fp32 accumulation",
            "seconds": dt, "gbps": b / dt / 1e9, "tokens_per_s_whole_model": n_seqs / (dt * L),
            "max_abs_err_vs_c_oracle": float(np.abs(o.numpy() - want[:n_seqs]).max()),
        }
    return res


PROGRESS = False


def progress(dist, what):
    if PROGRESS:
        print(f"# [rank {dist.rank}] {what}", file=sys.stderr, flush=True)


def main(argv=None):
    args = parse_args(argv)
    if args.cpu_plumbing_test:
        return cpu_plumbing_main(args)
    global PROGRESS
    PROGRESS = args.progress

    import numpy as np
    import torch
    import __graft_entry__

    H, D, L_model, model_name = WORKLOADS[args.workload]
    L = args.layers or L_model
    S, CTX, B = args.batch, args.ctx, args.block_size
    tdtype = {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32}[args.dtype]

    dist = Dist(args.dist_backend)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    dev_index = dist.local_rank % torch.cuda.device_count()  # one rank per GPU; ranks wrap only in a gloo rehearsal
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    pkg = __graft_entry__.load_package()
    attention_ops, cache_ops = pkg.attention_ops, pkg.cache_ops
    torch.manual_seed(0)

    if args.tp or args.tp_emulate:
        # ---- the reference's tensor parallelism: heads sharded, o_proj all-reduce in the step ----
        assert not (args.tp_emulate and dist.active), "--tp-emulate is a one-GPU mode"
        elapsed, info, w = tp_leg(torch, dist, device, pkg, args, H, D, L, tdtype, model_name, emulate=args.tp_emulate)
        n_tp = info["tp"]
        if dist.rank == 0:
            config = {
                "workload": f"{model_name} decode hot path, heads sharded over {n_tp} GPUs"
                            f"{' (EMULATED on one GPU, see tp.emulated)' if args.tp_emulate else ''}: {info['layers']} layers x "
                            f"({info['step']}), batch {S}, {info['heads_per_rank']} of {H} heads x {D} per GPU, "
                            f"block {B}, ctx {CTX}, distinct physical blocks",
                "batch": S, "num_heads": H, "heads_per_gpu": info["heads_per_rank"], "head_size": D, "block_size": B,
                "context_len": CTX, "layers": info["layers"], "sharding": "heads/tp", "parallelism": f"tp{n_tp}",
            }
            extra = {"tp": {k: v for k, v in info.items() if k not in ("value", "unit", "ms_per_step", "step")}}
            extra["ranks"] = {k: info[k] for k in ("backend", "world", "devices_seen", "rccl_version") if k in info}
            print(json.dumps(result_line(args, dist.world, elapsed, float(S * args.steps), "tokens/s", config, extra,
                                         metric_text(args, model_name, n_tp), "strong")))
        dist.close()
        return

    # ---- synthetic inputs, resident in HBM before timing ----------------------------------
    w = Workload(torch, device, H, D, L, S, CTX, B, tdtype, rank=dist.rank)
    L = w.L
    # The layers of a step share context_lens, hence the launch plan that depends on them
    # alone: layer 0 makes it into the step's workspace, the others reuse it (what
    # PagedAttention.forward of this package does through the step's InputMetadata).
    ws_bytes = attention_ops.decode_workspace_bytes(S, H, D, B, CTX)
    workspace = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=device)

    if args.roofline_only:
        # Bring the GPU to the state the full bench measures in — half a second of the same
        # launches over every layer's caches — before timing: the full bench has its warm-up and
        # timed steps before the isolated launches; this mode's first launches would otherwise be
        # timed during the clock ramp and the first sweep over 128 GiB (650 instead of 633 us).
        # A kernel trace of this command then holds several hundred launches in the steady state
        # and a few dozen before it: its per-dispatch median is the steady state's.
        attention_ops.single_query_cached_kv_attention_planned(
            w.out, w.query, w.key_caches[0], w.value_caches[0], w.scale, w.block_tables, w.context_lens, B, CTX,
            workspace, False, w.total_ctx)
        t_end = time.perf_counter() + 0.5
        while time.perf_counter() < t_end:
            for l in range(L):
                attention_ops.single_query_cached_kv_attention_planned(
                    w.out, w.query, w.key_caches[l], w.value_caches[l], w.scale, w.block_tables, w.context_lens, B, CTX,
                    workspace, True, w.total_ctx)
            torch.cuda.synchronize()
        roof = decode_roofline(torch, attention_ops, cache_ops, w, workspace, 40)
        if dist.rank == 0:
            print(json.dumps({"roofline": roof}))
        dist.close()
        return

    def step():
        for l in range(L):
            cache_ops.reshape_and_cache(w.key, w.value, w.key_caches[l], w.value_caches[l], w.slot_mapping)
            attention_ops.single_query_cached_kv_attention_planned(
                w.out, w.query, w.key_caches[l], w.value_caches[l], w.scale, w.block_tables, w.context_lens, B, CTX,
                workspace, l > 0, w.total_ctx)

    progress(dist, "inputs resident; timing the replica step")
    elapsed = timed_region(dist, device, step, args.steps, args.warmup, torch.cuda.synchronize)
    units = dist.sum_float(float(S * args.steps), device)
    progress(dist, f"replica step timed: {elapsed * 1e3 / args.steps:.3f} ms")

    extra = {}
    if dist.active:
        extra["ranks"] = dist.describe(torch)  # collective: every rank calls it
    # ---- roofline of the dominant kernel (decode attention) ------------------------------------
    if not args.no_roofline and dist.rank == 0:
        roof = decode_roofline(torch, attention_ops, cache_ops, w, workspace, max(2 * L, 20))
        roof["kernel"] = "pa_decode_kernel (single_query_cached_kv_attention)"
        roof["traffic"], roof["traffic_source"] = pmc_traffic("pa_decode_kernel", args)
        # The same kernel inside the step, i.e. right after reshape_and_cache has written this
        # layer's new K/V; a kernel trace of the whole command averages over both kinds.
        evs2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2 * L)]
        for i, (e0, e1) in enumerate(evs2):
            l = i % L
            cache_ops.reshape_and_cache(w.key, w.value, w.key_caches[l], w.value_caches[l], w.slot_mapping)
            e0.record()
            attention_ops.single_query_cached_kv_attention_planned(
                w.out, w.query, w.key_caches[l], w.value_caches[l], w.scale, w.block_tables, w.context_lens, B, CTX,
                workspace, True, w.total_ctx)
            e1.record()
        torch.cuda.synchronize()
        in_step = sorted(e0.elapsed_time(e1) for e0, e1 in evs2)
        roof["in_step_launch_ms_median"] = median(in_step)
        roof["in_step_achieved"] = roof["bytes_per_launch"] / (median(in_step) * 1e-3) / 1e9
        # What this box's memory does for a plain device-to-device copy, next to the vendor peak:
        # one layer's value cache over another's, read + written bytes counted.
        if L >= 2:
            src, dst = w.value_caches[L - 1], w.value_caches[L - 2]  # synthetic values either way

            def copy(i):
                dst.copy_(src)
            ts = time_launches(torch, copy, 10, warmup=2)
            nbytes = 2 * src.numel() * src.element_size()
            roof["device_copy"] = {"unit": "GB/s", "achieved": nbytes / (median(ts) * 1e-3) / 1e9,
                                   "bytes": nbytes, "what": "torch tensor copy on the device, read + written bytes"}
        extra["roofline"] = roof

    # ---- the microbenchmark shape of BASELINE.json configs[1], same measurement -----------------
    if not args.no_c2 and not args.no_roofline and dist.rank == 0 and args.workload != "opt125m_c2":
        H2, D2, _, name2 = WORKLOADS["opt125m_c2"]
        w2 = Workload(torch, device, H2, D2, 4, 256, 1024, 16, torch.float16, free_fraction=0.5)
        ws2 = torch.empty(max(attention_ops.decode_workspace_bytes(256, H2, D2, 16, 1024), 8), dtype=torch.uint8,
                          device=device)
        c2 = decode_roofline(torch, attention_ops, cache_ops, w2, ws2, 40)
        c2["workload"] = ("single_query_cached_kv_attention microbench: bs 256, 12 heads, head_size 64, block_size 16, "
                          "ctx 1024, fp16, distinct physical blocks, 4 layers' caches walked in turn")
        extra["c2"] = c2
        del w2, ws2

    def emit():
        if dist.rank == 0:
            config = {
                "workload": f"{model_name} decode hot path: {L} layers x (reshape_and_cache + single_query_cached_kv_attention), "
                            f"batch {S} per GPU, {H} heads x {D}, block {B}, ctx {CTX}, "
                            f"distinct physical blocks, per-layer 2x{w.cache_bytes / 2 / 2**30:.2f} GiB caches",
                "batch_per_gpu": S, "num_heads": H, "head_size": D, "block_size": B, "context_len": CTX,
                "layers": L, "sharding": "independent replicas per GPU (no data-path collective); "
                                         "the head-sharded form is in \"tp\" (N > 1) or under --tp",
            }
            print(json.dumps(result_line(args, dist.world, elapsed, units, "tokens/s", config, extra,
                                         metric_text(args, model_name, 1))), flush=True)

    # ---- N > 1: the head-sharded form measured in the same process ---------------------------------
    if dist.active and not args.no_tp_leg and H % dist.world == 0:
        # An extra measurement: whatever goes wrong in it must not cost the line above.  Errors are
        # caught below; against a HANG (a collective that some rank never enters) every rank runs a
        # deadline: when it passes, rank 0 prints the line with what it has and all ranks leave.
        line_lock = threading.Lock()  # held by whoever prints the line

        def expire():
            if line_lock.acquire(blocking=False):
                for leg in ("tp", "tp_push"):
                    if leg == "tp" or args.tp_collective != "push":
                        extra.setdefault(leg, {"error": f"did not finish within the {args.tp_leg_deadline:.0f} s deadline of the extra legs"})
                # every rank leaves with status 0 (a rank failing first would have the launcher end rank 0
                # before its line is out): the line itself says that a leg hung
                extra["extra_legs_deadline_expired"] = True
                emit()
            if CHILD is not None:
                CHILD.kill()
            os._exit(0)

        t_legs = time.monotonic()
        deadline = threading.Timer(args.tp_leg_deadline, expire)
        deadline.daemon = True
        deadline.start()
        progress(dist, "head-sharded leg")
        try:
            tp_elapsed, info, _ = tp_leg(torch, dist, device, pkg, args, H, D, L, tdtype, model_name, shared_caches=w)
            extra["tp"] = info
        except Exception as e:  # noqa: BLE001
            extra["tp"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        if args.tp_collective != "push":
            # the same step with this package's own all-reduce over peer-mapped inboxes
            progress(dist, "head-sharded leg, one-shot push all-reduce (child processes)")
            try:
                info = push_leg_in_children(args, dist, device, t_legs + args.tp_leg_deadline - 15.0 - time.monotonic())
                if info is not None:
                    extra["tp_push"] = info
            except Exception as e:  # noqa: BLE001
                extra["tp_push"] = {"error": f"{type(e).__name__}: {e}"[:500]}
        if not line_lock.acquire(blocking=False):
            time.sleep(3600)  # the deadline passed at this very moment: its thread prints and ends the process
        deadline.cancel()

    # ---- the operators either side of the decode kernel (N = 1, default workload) --------------------
    default_workload = (args.workload, args.dtype, args.block_size) == ("llama7b", "float16", 16)
    if not args.no_secondary and not args.no_roofline and dist.world == 1 and default_workload:
        progress(dist, "secondary measurements")
        guarded(extra, "cache_write", lambda: cache_write_measurement(torch, cache_ops, args, w))
        guarded(extra, "small_batch", lambda: small_batch_measurement(torch, attention_ops, w))
        guarded(extra, "prompt", lambda: prompt_measurement(torch, attention_ops, device, H, D))
        guarded(extra, "decoder_step", lambda: decoder_step_measurement(pkg, w))

    # ---- CPU baseline --------------------------------------------------------------------------------
    if not args.no_cpu_baseline and dist.rank == 0 and dist.world == 1:
        extra["cpu_baseline"] = cpu_baseline(torch, np, args, w, L, attention_ops, cache_ops)

    emit()
    dist.close()


if __name__ == "__main__":
    main()
