#!/usr/bin/env python3
"""Offline-throughput benchmark in the form of the reference's
benchmarks/benchmark_throughput.py, with synthetic requests (no dataset, no
tokenizer, no checkpoint can be fetched here).

  * requests: --num-prompts (prompt_len, output_len) pairs drawn with a fixed
    seed from log-normal lengths and passed through the script's own filter
    (benchmark_throughput.py:57-67: both >= 4, prompt <= 1024, sum <= 2048);
    prompt token ids are random; every request generates exactly output_len
    tokens (ignore_eos=True, max_tokens=output_len, :90-97), sampled at
    temperature 1.0, top_p 1.0;
  * model: a LLaMA-shaped decoder with random weights (normal, std 0.02);
    the GEMMs are torch's (hipBLASLt), everything between them is this
    package: rms_norm, rotary_embedding_neox, multi_query_kv_attention,
    reshape_and_cache, single_query_cached_kv_attention, silu_and_mul;
  * batching: the reference scheduler's policy restated
    (vllm/core/scheduler.py:104-206): every step carries one token of every
    running sequence plus as many waiting prompts, first come first served, as
    fit in max_num_batched_tokens (2560) and max_num_seqs (256) and in the
    free cache blocks; finished sequences free their blocks at once.  With
    the default cache size preemption never triggers; with a small cache
    (--num-blocks) and a CPU swap space (--num-cpu-blocks) the latest-arrived
    running group is preempted by swapping (scheduler.py:340-417): its blocks go
    to pinned host memory through CacheEngine.swap_out on the cache stream,
    and come back through swap_in, ahead of any new prompt, once blocks are free;
  * parallel sampling (--n 3, the serving benchmark's second setting,
    benchmarks/benchmark_serving.py:105-159): after its prompt step a request
    forks into n sequences that share the prompt's blocks (reference counts,
    vllm/core/block_manager.py:111-135); the first token a sequence appends to a
    shared last block copies that block (copy-on-write, CacheEngine.copy ->
    cache_ops.copy_blocks), and neighbours in the batch that still share full
    blocks meet in the decode kernel's L2;
  * per-GPU shards (--model 13b-tp2 | neox20b-tp4, or --heads-per-rank /
    --head-size / --rotary-dim): one rank's share of a tensor-parallel model —
    its heads, its slices of the projections — on one GPU, without the
    all-reduce (there is no second GPU to reduce with; the result says so);
  * result: the script's own figures, requests/s and (prompt + output)
    tokens/s over the wall time of the whole run (:186-191), and output
    tokens/s.

    python tools/bench_throughput.py [--model 7b|13b|13b-tp2|neox20b-tp4|tiny] [--n 1|3] [--num-prompts 1000]
                                     [--num-blocks N --num-cpu-blocks M] [--json out.json]

Device time per operator class: run the same command under
`rocprofv3 --kernel-trace --stats --output-format csv` and give the
*_kernel_stats.csv to tools/summarize_kernel_classes.py; the result's
paged_attention_kv_bytes over the pa_decode_* time is the bandwidth the paged
attention reached on these batches.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from collections import deque
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

MODELS = {
    "7b": dict(hidden=4096, inter=11008, heads=32, layers=32, vocab=32000),
    "13b": dict(hidden=5120, inter=13824, heads=40, layers=40, vocab=32000),
    "tiny": dict(hidden=256, inter=512, heads=4, layers=2, vocab=500),
    # one rank's share of a tensor-parallel model: heads / tp, inter / tp, vocab / tp (the logits' shard)
    "13b-tp2": dict(hidden=5120, inter=13824 // 2, heads=20, head_size=128, layers=40, vocab=32000 // 2,
                    what="LLaMA-13B, rank 0 of tensor-parallel 2"),
    "neox20b-tp4": dict(hidden=6144, inter=24576 // 4, heads=16, head_size=96, rotary_dim=24, layers=44, vocab=50432 // 4,
                        mlp="gelu", norm="layernorm", what="GPT-NeoX-20B, rank 0 of tensor-parallel 4"),
    "tiny-neox": dict(hidden=192, inter=256, heads=2, head_size=96, rotary_dim=24, layers=2, vocab=300, mlp="gelu",
                      norm="layernorm", what="a two-layer model of the GPT-NeoX shard's head shape"),
}


def sample_requests(num, seed, max_prompt=1024, max_total=2048):
    """(prompt_len, output_len) pairs: log-normal lengths (medians 160 and 140
    tokens, sigma 1.0 — long-tailed like a chat corpus), then the reference
    script's filter."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < num:
        p = int(rng.lognormal(np.log(160.0), 1.0))
        o = int(rng.lognormal(np.log(140.0), 1.0))
        if p < 4 or o < 4:
            continue
        if p > max_prompt or p + o > max_total:
            continue
        out.append((p, o))
    return out


class Sequence:
    __slots__ = ("tokens", "prompt_len", "output_len", "table", "margins")

    def __init__(self, prompt_tokens, output_len):
        self.tokens = list(prompt_tokens)
        self.prompt_len = len(self.tokens)
        self.output_len = output_len
        self.table = []
        self.margins = []  # record_margins: top-1 minus top-2 logit of every token this sequence appended

    def get_len(self):
        return len(self.tokens)

    def get_token_ids(self):
        return self.tokens

    def get_last_token_id(self):
        return self.tokens[-1]

    def finished(self):
        return len(self.tokens) - self.prompt_len >= self.output_len

    def fork(self):
        child = Sequence(self.tokens, self.output_len)
        child.prompt_len = self.prompt_len
        child.table = list(self.table)
        return child


class Group:
    """A request: one sequence until its prompt has run, n sequences after (vllm/sequence.py SequenceGroup)."""
    __slots__ = ("seqs", "n", "arrival", "done")

    def __init__(self, seq, n, arrival):
        self.seqs = [seq]
        self.n = n
        self.arrival = arrival
        self.done = []


class BlockPool:
    """Reference-counted GPU blocks and a pool of CPU blocks (vllm/core/block_manager.py:10-60, 111-160)."""

    def __init__(self, num_gpu, num_cpu):
        self.free_gpu = list(range(num_gpu - 1, -1, -1))
        self.free_cpu = list(range(num_cpu - 1, -1, -1))
        self.ref = {}       # GPU block -> number of sequences whose table holds it
        self.cpu_ref = {}   # the same for swapped-out blocks

    def alloc(self):
        b = self.free_gpu.pop()
        self.ref[b] = 1
        return b

    def share(self, table):
        for b in table:
            self.ref[b] += 1

    def release(self, table):
        for b in table:
            self.ref[b] -= 1
            if self.ref[b] == 0:
                del self.ref[b]
                self.free_gpu.append(b)

    def swap_out(self, seqs):
        """Moves the GPU blocks of these sequences to CPU blocks; returns {gpu: cpu}.  A block shared by
        several of the sequences moves once (block_manager.py:196-224)."""
        mapping = {}
        for s in seqs:
            for i, b in enumerate(s.table):
                if b not in mapping:
                    mapping[b] = self.free_cpu.pop()
                    self.cpu_ref[mapping[b]] = 0
                self.cpu_ref[mapping[b]] += 1
                self.ref[b] -= 1
                if self.ref[b] == 0:
                    del self.ref[b]
                    self.free_gpu.append(b)
                s.table[i] = mapping[b]
        return mapping

    def swap_in(self, seqs):
        mapping = {}
        for s in seqs:
            for i, c in enumerate(s.table):
                if c not in mapping:
                    mapping[c] = self.free_gpu.pop()
                    self.ref[mapping[c]] = 0
                self.ref[mapping[c]] += 1
                self.cpu_ref[c] -= 1
                if self.cpu_ref[c] == 0:
                    del self.cpu_ref[c]
                    self.free_cpu.append(c)
                s.table[i] = mapping[c]
        return mapping


def build_model(pkg, cfg, dtype, device, block_size, num_blocks, max_position, num_cpu_blocks=1):
    import torch
    import torch.nn.functional as F
    from vllm_amd.layers import PagedAttentionWithRoPE, RMSNorm, SiluAndMul
    from vllm_amd.worker import CacheEngine, simple_configs

    hidden, inter, H, L, vocab = cfg["hidden"], cfg["inter"], cfg["heads"], cfg["layers"], cfg["vocab"]
    D = cfg.get("head_size", hidden // H)
    local = H * D  # this rank's slice of the attention width (== hidden for an unsharded model)
    gelu, layernorm = cfg.get("mlp") == "gelu", cfg.get("norm") == "layernorm"
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        def w(*shape):
            return torch.empty(*shape, dtype=dtype, device=device).normal_(0, cfg.get("weight_std", 0.02))

        def norm():
            if layernorm:  # GPT-NeoX: torch's layer norm (not one of this package's operators)
                g, b_ = torch.ones(hidden, dtype=dtype, device=device), torch.zeros(hidden, dtype=dtype, device=device)
                return lambda x: F.layer_norm(x, (hidden,), g, b_, 1e-5)
            return RMSNorm(hidden).to(device)

        embed, lm_head = w(vocab, hidden), w(hidden, vocab)
        layers = [dict(ln1=norm(), ln2=norm(), qkv=w(hidden, 3 * local), o=w(local, hidden),
                       up=w(hidden, inter if gelu else 2 * inter), down=w(inter, hidden)) for _ in range(L)]
        final_norm = norm()
        attn = PagedAttentionWithRoPE(H, D, D ** -0.5, rotary_dim=cfg.get("rotary_dim", D), max_position=max_position).to(device)
    finally:
        torch.set_default_dtype(old)
    act = SiluAndMul()
    cc, mc, pc = simple_configs(L, H, D, dtype, block_size, num_gpu_blocks=num_blocks, num_cpu_blocks=max(num_cpu_blocks, 1))
    engine = CacheEngine(cc, mc, pc)

    def forward(tokens, positions, meta, rows, cache_events=None):
        """Hidden states of `rows` (the positions that are sampled from) -> logits."""
        x = embed[tokens]
        for i, ly in enumerate(layers):
            kc, vc = engine.gpu_cache[i]
            ev = cache_events[i] if cache_events is not None else None
            q, k, v = (ly["ln1"](x) @ ly["qkv"]).chunk(3, dim=-1)
            a = attn(positions, q, k, v, kc, vc, meta, ev) @ ly["o"]
            if gelu:  # GPT-NeoX: attention and MLP both read the layer's input (parallel residual)
                x = x + a + F.gelu(ly["ln2"](x) @ ly["up"]) @ ly["down"]
            else:
                x = x + a
                x = x + act(ly["ln2"](x) @ ly["up"]) @ ly["down"]
        return final_norm(x[rows]) @ lm_head

    return forward, engine


def run(model="7b", num_prompts=1000, seed=0, max_num_batched_tokens=2560, max_num_seqs=256, block_size=16,
        num_blocks=0, dtype_name="float16", device="cuda:0", verbose=True, step_log_path="", n=1, num_cpu_blocks=0,
        sampling="random", requests=None, cfg_overrides=None, record_margins=False):
    """sampling: "random" = temperature 1.0 (the benchmark's setting); "greedy_distinct" = a request's j-th
    sequence starts with the j-th most likely token after the prompt and continues greedily (deterministic:
    for the tests).  requests: [(prompt token ids, output_len)] instead of the sampled ones.
    Returns the result dict; with requests given, result["sequences"] lists every finished sequence as
    (request index, tokens, margins)."""
    import torch
    import __graft_entry__

    pkg = __graft_entry__.load_package()  # registers the package as vllm_amd
    from vllm_amd.worker import prepare_inputs
    cfg = dict(MODELS[model])
    cfg.update(cfg_overrides or {})
    dtype = getattr(torch, dtype_name)
    B = block_size
    L, H = cfg["layers"], cfg["heads"]
    D = cfg.get("head_size", cfg["hidden"] // H)
    es = torch.tensor([], dtype=dtype).element_size()
    block_bytes = 2 * L * H * D * B * es
    if num_blocks <= 0:
        # every running sequence at the longest total length, plus slack: preemption never needed
        num_blocks = max_num_seqs * (2048 // B) + 1024
        free_b, _ = torch.cuda.mem_get_info(torch.device(device))
        weights_b = es * (L * (4 * cfg["hidden"] * H * D + 3 * cfg["hidden"] * cfg["inter"]) + 2 * cfg["vocab"] * cfg["hidden"])
        num_blocks = int(min(num_blocks, max(64, (0.9 * free_b - weights_b) // block_bytes)))
    torch.manual_seed(seed)
    forward, engine = build_model(pkg, cfg, dtype, device, B, num_blocks, max_position=2048, num_cpu_blocks=num_cpu_blocks)
    rng = np.random.default_rng(seed + 1)
    given = requests is not None
    if not given:
        lens = sample_requests(num_prompts, seed)
        requests = [(rng.integers(0, cfg["vocab"], size=p).tolist(), o) for p, o in lens]
    waiting = deque(Group(Sequence(p, o), n, i) for i, (p, o) in enumerate(requests))
    pool = BlockPool(num_blocks, num_cpu_blocks)
    watermark = int(0.01 * num_blocks)  # block_manager.py: keep 1 % free when admitting prompts
    running, swapped, finished = [], [], []
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)

    steps = prompt_steps = 0
    stats = dict(copy_on_write_blocks=0, copy_blocks_calls=0, swap_out_groups=0, swap_out_blocks=0, swap_in_groups=0,
                 swap_in_blocks=0, forks=0)
    step_log = []
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    while waiting or running or swapped:
        to_copy, swap_out_map, swap_in_map = {}, {}, {}
        # 1. a slot for the token every running sequence sampled last step (scheduler.py:141-160): a new block,
        #    or — appending to a block other sequences still share — a private copy of it.  When the free
        #    blocks do not cover that, the group that arrived last is preempted: swapped out, or, without
        #    swap space, the run ends (this harness does not recompute)
        running.sort(key=lambda g: g.arrival)
        kept = []
        while running:
            g = running.pop(0)
            need = sum(1 for s in g.seqs if len(s.table) * B < s.get_len() or pool.ref[s.table[-1]] > 1)
            while len(pool.free_gpu) < need and running:
                victim = running.pop()  # lowest priority: the latest arrival
                _preempt(victim, pool, swap_out_map, swapped, stats)
            if len(pool.free_gpu) < need:
                _preempt(g, pool, swap_out_map, swapped, stats)
                continue
            for s in g.seqs:
                if len(s.table) * B < s.get_len():
                    s.table.append(pool.alloc())
                elif pool.ref[s.table[-1]] > 1:  # copy-on-write (block_manager.py:126-135)
                    src = s.table[-1]
                    pool.ref[src] -= 1
                    s.table[-1] = pool.alloc()
                    to_copy.setdefault(src, []).append(s.table[-1])
                    stats["copy_on_write_blocks"] += 1
            kept.append(g)
        running = kept
        # 2. swapped groups come back first, oldest first, while blocks allow (scheduler.py:162-185)
        swapped.sort(key=lambda g: g.arrival)
        while swapped and not swap_out_map:
            g = swapped[0]
            blocks = len({c for s in g.seqs for c in s.table})
            if len(pool.free_gpu) - blocks - len(g.seqs) < watermark:
                break
            if sum(len(x.seqs) for x in running) + len(g.seqs) > max_num_seqs:
                break
            swapped.pop(0)
            swap_in_map.update(pool.swap_in(g.seqs))
            for s in g.seqs:  # the slot of the token it sampled before it was preempted
                if len(s.table) * B < s.get_len():
                    s.table.append(pool.alloc())
                elif pool.ref[s.table[-1]] > 1:
                    src = s.table[-1]
                    pool.ref[src] -= 1
                    s.table[-1] = pool.alloc()
                    to_copy.setdefault(src, []).append(s.table[-1])
                    stats["copy_on_write_blocks"] += 1
            stats["swap_in_groups"] += 1
            running.append(g)
        stats["swap_in_blocks"] += len(swap_in_map)
        # 3. waiting prompts, first come first served — none while anything is swapped out (scheduler.py:187-190)
        run_seqs = [s for g in running for s in g.seqs]
        batched = len(run_seqs)
        admitted = []
        while waiting and not swapped:
            g = waiting[0]
            s = g.seqs[0]
            need = (s.get_len() + B - 1) // B
            if len(pool.free_gpu) - need < watermark:
                break
            if batched + s.get_len() > max_num_batched_tokens:
                break
            if len(run_seqs) + sum(x.n for x in admitted) + g.n > max_num_seqs:
                break
            waiting.popleft()
            s.table = [pool.alloc() for _ in range(need)]
            admitted.append(g)
            batched += s.get_len()
        if not admitted and not run_seqs:
            if swap_out_map:  # everything that ran was preempted this step: move it out, then bring the oldest back
                engine.swap_out(swap_out_map)
                continue
            if swapped or waiting:
                raise RuntimeError("nothing can run: the cache is too small for the next group (raise --num-blocks)")
            break
        # 4. cache operations, then the model (worker.py:239-283)
        issued = False
        if swap_in_map:
            engine.swap_in(swap_in_map)
            issued = True
        if swap_out_map:
            engine.swap_out(swap_out_map)
            issued = True
        if to_copy:
            engine.copy(to_copy)
            stats["copy_blocks_calls"] += 1
        prompts = [g.seqs[0] for g in admitted]
        order = prompts + run_seqs
        groups = [SimpleNamespace(request_id=str(i), is_prompt=i < len(prompts), seq_data={i: s}, sampling_params=None,
                                  block_tables={i: s.table}) for i, s in enumerate(order)]
        gen_ctx_tokens = sum(s.get_len() for s in run_seqs)  # context tokens the paged attention reads this step
        longest = max((s.get_len() for s in run_seqs), default=0)
        t0 = time.perf_counter()
        tokens, positions, meta = prepare_inputs(groups, B, device=device)
        ends = np.cumsum([s.get_len() for s in prompts]) - 1 if prompts else np.zeros(0, dtype=np.int64)
        n_prompt_tokens = int(ends[-1]) + 1 if prompts else 0
        rows = np.concatenate([ends, n_prompt_tokens + np.arange(len(run_seqs))]).astype(np.int64)
        rows_t = torch.from_numpy(rows).to(device, non_blocking=True)
        logits = forward(tokens, positions, meta, rows_t, engine.events if issued else None)
        # 5. sample: n tokens from a prompt's distribution (the request forks), one per running sequence
        logits = logits.float()
        top = None
        if sampling == "random":
            probs = torch.softmax(logits, dim=-1)
            draws = torch.multinomial(probs, max(n, 1), replacement=True, generator=gen).tolist()  # host sync, as the engine's
        else:
            top = torch.topk(logits, max(n, 2), dim=-1)
            draws = top.indices.tolist()
        margins = (top.values[:, 0] - top.values[:, 1]).tolist() if (record_margins and top is not None) else None
        for i, g in enumerate(admitted):
            parent = g.seqs[0]
            children = [parent] + [parent.fork() for _ in range(g.n - 1)]
            for c in children[1:]:
                pool.share(c.table)
            stats["forks"] += g.n - 1
            for j, c in enumerate(children):
                c.tokens.append(int(draws[i][j]))
                if margins is not None:
                    c.margins.append(margins[i] if j == 0 else float("inf"))  # the j-th best token is a choice, not a tie
            g.seqs = children
        for i, s in enumerate(run_seqs):
            s.tokens.append(int(draws[len(prompts) + i][0]))
            if margins is not None:
                s.margins.append(margins[len(prompts) + i])
        still = []
        for g in admitted + running:
            alive = []
            for s in g.seqs:
                if s.finished():
                    pool.release(s.table)
                    s.table = []
                    g.done.append(s)
                else:
                    alive.append(s)
            g.seqs = alive
            if alive:
                still.append(g)
            else:
                finished.append(g)
        running = still
        steps += 1
        prompt_steps += bool(admitted)
        step_log.append((len(prompts), n_prompt_tokens, len(run_seqs), time.perf_counter() - t0, gen_ctx_tokens, longest))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    assert not pool.ref and not pool.cpu_ref and len(pool.free_gpu) == num_blocks, "blocks leaked"

    total_prompt = sum(len(p) for p, _ in requests)
    total_output = sum(o for _, o in requests) * n
    decode_only = [w[3] for w in step_log if w[0] == 0 and w[2] >= max_num_seqs - 8]
    result = {
        "benchmark": "benchmark_throughput.py restated with synthetic requests",
        "model": f"{cfg.get('what', 'LLaMA-' + model)} shape, random weights, {dtype_name}",
        "heads": H, "head_size": D, "rotary_dim": cfg.get("rotary_dim", D), "layers": L,
        "all_reduce": "none: one rank's share on one GPU" if "what" in cfg and "rank" in cfg["what"] else "not needed (unsharded)",
        "sequences_per_request": n, "sampling": sampling,
        "num_requests": len(requests), "prompt_tokens": total_prompt, "output_tokens": total_output,
        "mean_prompt_len": total_prompt / len(requests), "mean_output_len": total_output / len(requests) / n,
        "max_num_batched_tokens": max_num_batched_tokens, "max_num_seqs": max_num_seqs, "block_size": B,
        "num_gpu_blocks": num_blocks, "num_cpu_blocks": num_cpu_blocks, "steps": steps, "steps_with_prompts": prompt_steps,
        "elapsed_s": elapsed,
        "requests_per_s": len(requests) / elapsed,
        "total_tokens_per_s": (total_prompt + total_output) / elapsed,
        "output_tokens_per_s": total_output / elapsed,
        "full_batch_decode_step_ms": 1e3 * float(np.median(decode_only)) if decode_only else None,
        # what the paged attention had to read over the whole run: K and V of every context token, every layer
        "context_tokens_read": int(sum(w[4] for w in step_log)),
        "paged_attention_kv_bytes": int(sum(w[4] for w in step_log)) * 2 * H * D * L * es,
        **stats,
    }
    if given:
        result["sequences"] = [(g.arrival, s.tokens, s.margins) for g in finished for s in g.done]
    if step_log_path:
        with open(step_log_path, "w") as f:  # per step: generation sequences, their context tokens, the longest context
            json.dump([(w[2], w[4], w[5]) for w in step_log], f)
    if verbose:
        print(json.dumps({k: v for k, v in result.items() if k != "sequences"}, indent=1))
    return result


def _preempt(group, pool, swap_out_map, swapped, stats):
    """Preemption by swapping (scheduler.py:340-417; groups of several sequences are always swapped, :352-360)."""
    blocks = len({b for s in group.seqs for b in s.table})
    if len(pool.free_cpu) < blocks:
        raise RuntimeError("out of cache blocks and of swap space: raise --num-blocks or --num-cpu-blocks "
                           "(the reference aborts here too, scheduler.py:408-413)")
    swap_out_map.update(pool.swap_out(group.seqs))
    stats["swap_out_groups"] += 1
    stats["swap_out_blocks"] += blocks
    swapped.append(group)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b", choices=sorted(MODELS))
    ap.add_argument("--num-prompts", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-num-batched-tokens", type=int, default=2560)
    ap.add_argument("--max-num-seqs", type=int, default=256)
    ap.add_argument("--block-size", type=int, default=16)
    ap.add_argument("--num-blocks", type=int, default=0)
    ap.add_argument("--num-cpu-blocks", type=int, default=0, help="swap space, in blocks (0: no preemption possible)")
    ap.add_argument("--n", type=int, default=1, help="sequences per request (parallel sampling)")
    ap.add_argument("--heads-per-rank", type=int, default=0, help="override: attention heads on this GPU")
    ap.add_argument("--head-size", type=int, default=0)
    ap.add_argument("--rotary-dim", type=int, default=0)
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--dtype", default="float16")
    ap.add_argument("--json", default="")
    ap.add_argument("--step-log", default="", help="write (sequences, context tokens, longest context) of every step here")
    args = ap.parse_args()
    over = {}
    if args.heads_per_rank:
        over["heads"] = args.heads_per_rank
    if args.head_size:
        over["head_size"] = args.head_size
    if args.rotary_dim:
        over["rotary_dim"] = args.rotary_dim
    if args.layers:
        over["layers"] = args.layers
    res = run(args.model, args.num_prompts, args.seed, args.max_num_batched_tokens, args.max_num_seqs,
              args.block_size, args.num_blocks, args.dtype, step_log_path=args.step_log, n=args.n,
              num_cpu_blocks=args.num_cpu_blocks, cfg_overrides=over)
    if args.json:
        with open(args.json, "w") as f:
            json.dump({k: v for k, v in res.items() if k != "sequences"}, f, indent=1)


if __name__ == "__main__":
    main()
