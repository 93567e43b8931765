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
    free cache blocks; finished sequences free their blocks at once.  The
    cache is sized so that preemption never triggers (the run aborts if it
    would), which keeps swapping and recomputation out of the picture;
  * result: the script's own figures, requests/s and (prompt + output)
    tokens/s over the wall time of the whole run (:186-191), and output
    tokens/s.

    python tools/bench_throughput.py [--model 7b|13b|tiny] [--num-prompts 1000] [--json out.json]

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
    __slots__ = ("tokens", "prompt_len", "output_len", "table")

    def __init__(self, prompt_tokens, output_len):
        self.tokens = list(prompt_tokens)
        self.prompt_len = len(self.tokens)
        self.output_len = output_len
        self.table = []

    def get_len(self):
        return len(self.tokens)

    def get_token_ids(self):
        return self.tokens

    def get_last_token_id(self):
        return self.tokens[-1]

    def finished(self):
        return len(self.tokens) - self.prompt_len >= self.output_len


def build_model(pkg, cfg, dtype, device, block_size, num_blocks, max_position):
    import torch
    from vllm_amd.layers import PagedAttentionWithRoPE, RMSNorm, SiluAndMul
    from vllm_amd.worker import CacheEngine, simple_configs

    hidden, inter, H, L, vocab = cfg["hidden"], cfg["inter"], cfg["heads"], cfg["layers"], cfg["vocab"]
    D = hidden // H
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        def w(*shape):
            return torch.empty(*shape, dtype=dtype, device=device).normal_(0, 0.02)

        embed, lm_head = w(vocab, hidden), w(hidden, vocab)
        layers = [dict(ln1=RMSNorm(hidden).to(device), ln2=RMSNorm(hidden).to(device), qkv=w(hidden, 3 * hidden),
                       o=w(hidden, hidden), gate_up=w(hidden, 2 * inter), down=w(inter, hidden)) for _ in range(L)]
        final_norm = RMSNorm(hidden).to(device)
        attn = PagedAttentionWithRoPE(H, D, D ** -0.5, rotary_dim=D, max_position=max_position).to(device)
    finally:
        torch.set_default_dtype(old)
    act = SiluAndMul()
    cc, mc, pc = simple_configs(L, H, D, dtype, block_size, num_gpu_blocks=num_blocks, num_cpu_blocks=1)
    engine = CacheEngine(cc, mc, pc)

    def forward(tokens, positions, meta, rows):
        """Hidden states of `rows` (the positions that are sampled from) -> logits."""
        x = embed[tokens]
        for i, ly in enumerate(layers):
            kc, vc = engine.gpu_cache[i]
            q, k, v = (ly["ln1"](x) @ ly["qkv"]).chunk(3, dim=-1)
            x = x + attn(positions, q, k, v, kc, vc, meta, None) @ ly["o"]
            x = x + act(ly["ln2"](x) @ ly["gate_up"]) @ ly["down"]
        return final_norm(x[rows]) @ lm_head

    return forward


def run(model="7b", num_prompts=1000, seed=0, max_num_batched_tokens=2560, max_num_seqs=256, block_size=16,
        num_blocks=0, dtype_name="float16", device="cuda:0", verbose=True, step_log_path=""):
    import torch
    import __graft_entry__

    pkg = __graft_entry__.load_package()  # registers the package as vllm_amd
    from vllm_amd.worker import prepare_inputs
    cfg = MODELS[model]
    dtype = getattr(torch, dtype_name)
    B = block_size
    L, H, D = cfg["layers"], cfg["heads"], cfg["hidden"] // cfg["heads"]
    block_bytes = 2 * L * H * D * B * torch.tensor([], dtype=dtype).element_size()
    if num_blocks <= 0:
        # every running sequence at the longest total length, plus slack: preemption never needed
        num_blocks = max_num_seqs * (2048 // B) + 1024
        free_b, _ = torch.cuda.mem_get_info(torch.device(device))
        weights_b = 2 * (L * (4 * cfg["hidden"] ** 2 + 3 * cfg["hidden"] * cfg["inter"]) + 2 * cfg["vocab"] * cfg["hidden"])
        num_blocks = int(min(num_blocks, max(64, (0.9 * free_b - weights_b) // block_bytes)))
    torch.manual_seed(seed)
    forward = build_model(pkg, cfg, dtype, device, B, num_blocks, max_position=2048)
    rng = np.random.default_rng(seed + 1)
    requests = sample_requests(num_prompts, seed)
    waiting = deque(Sequence(rng.integers(0, cfg["vocab"], size=p).tolist(), o) for p, o in requests)
    free_blocks = list(range(num_blocks - 1, -1, -1))
    watermark = int(0.01 * num_blocks)  # block_manager.py: keep 1 % free when admitting prompts
    running = []
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)

    steps = prompt_steps = 0
    step_log = []
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    while waiting or running:
        # one new slot for every running sequence (its last sampled token is written this step)
        for s in running:
            if len(s.table) * B < s.get_len():
                if not free_blocks:
                    raise RuntimeError("out of cache blocks: this harness does not preempt; raise --num-blocks")
                s.table.append(free_blocks.pop())
        # admit waiting prompts, first come first served
        batched = len(running)
        admitted = []
        while waiting:
            s = waiting[0]
            need = (s.get_len() + B - 1) // B
            if len(free_blocks) - need < watermark:
                break
            if batched + s.get_len() > max_num_batched_tokens:
                break
            if len(running) + len(admitted) + 1 > max_num_seqs:
                break
            waiting.popleft()
            s.table = [free_blocks.pop() for _ in range(need)]
            admitted.append(s)
            batched += s.get_len()
        groups = [SimpleNamespace(request_id=str(i), is_prompt=i < len(admitted), seq_data={i: s}, sampling_params=None,
                                  block_tables={i: s.table}) for i, s in enumerate(admitted + running)]
        gen_ctx_tokens = sum(s.get_len() for s in running)  # context tokens the paged attention reads this step
        running_before = list(running)
        t0 = time.perf_counter()
        tokens, positions, meta = prepare_inputs(groups, B, device=device)
        ends = np.cumsum([s.get_len() for s in admitted]) - 1 if admitted else np.zeros(0, dtype=np.int64)
        n_prompt_tokens = int(ends[-1]) + 1 if admitted else 0
        rows = np.concatenate([ends, n_prompt_tokens + np.arange(len(running))]).astype(np.int64)
        rows_t = torch.from_numpy(rows).to(device, non_blocking=True)
        logits = forward(tokens, positions, meta, rows_t)
        probs = torch.softmax(logits.float(), dim=-1)
        next_tokens = torch.multinomial(probs, 1, generator=gen).squeeze(1).tolist()  # host sync, as the engine's
        order = admitted + running
        for s, tok in zip(order, next_tokens):
            s.tokens.append(int(tok))
        still = []
        for s in order:
            if s.finished():
                free_blocks.extend(s.table)
                s.table = []
            else:
                still.append(s)
        running = still
        steps += 1
        prompt_steps += bool(admitted)
        step_log.append((len(admitted), n_prompt_tokens, len(order) - len(admitted), time.perf_counter() - t0, gen_ctx_tokens,
                         max((s.get_len() for s in running_before), default=0)))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start

    total_prompt = sum(p for p, _ in requests)
    total_output = sum(o for _, o in requests)
    decode_only = [w[3] for w in step_log if w[0] == 0 and w[2] >= max_num_seqs - 8]
    result = {
        "benchmark": "benchmark_throughput.py restated with synthetic requests",
        "model": f"LLaMA-{model} shape, random weights, {dtype_name}",
        "num_requests": len(requests), "prompt_tokens": total_prompt, "output_tokens": total_output,
        "mean_prompt_len": total_prompt / len(requests), "mean_output_len": total_output / len(requests),
        "max_num_batched_tokens": max_num_batched_tokens, "max_num_seqs": max_num_seqs, "block_size": B,
        "num_gpu_blocks": num_blocks, "steps": steps, "steps_with_prompts": prompt_steps,
        "elapsed_s": elapsed,
        "requests_per_s": len(requests) / elapsed,
        "total_tokens_per_s": (total_prompt + total_output) / elapsed,
        "output_tokens_per_s": total_output / elapsed,
        "full_batch_decode_step_ms": 1e3 * float(np.median(decode_only)) if decode_only else None,
        # what the paged attention had to read over the whole run: K and V of every context token, every layer
        "context_tokens_read": int(sum(w[4] for w in step_log)),
        "paged_attention_kv_bytes": int(sum(w[4] for w in step_log)) * 2 * H * D * L
                                    * torch.tensor([], dtype=dtype).element_size(),
    }
    if step_log_path:
        with open(step_log_path, "w") as f:  # per step: generation sequences, their context tokens, the longest context
            json.dump([(w[2], w[4], w[5]) for w in step_log], f)
    if verbose:
        print(json.dumps(result, indent=1))
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b", choices=sorted(MODELS))
    ap.add_argument("--num-prompts", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-num-batched-tokens", type=int, default=2560)
    ap.add_argument("--max-num-seqs", type=int, default=256)
    ap.add_argument("--block-size", type=int, default=16)
    ap.add_argument("--num-blocks", type=int, default=0)
    ap.add_argument("--dtype", default="float16")
    ap.add_argument("--json", default="")
    ap.add_argument("--step-log", default="", help="write (sequences, context tokens, longest context) of every step here")
    args = ap.parse_args()
    res = run(args.model, args.num_prompts, args.seed, args.max_num_batched_tokens, args.max_num_seqs,
              args.block_size, args.num_blocks, args.dtype, step_log_path=args.step_log)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
