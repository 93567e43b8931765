#!/usr/bin/env python3
"""Whole LLaMA-shaped decode step around the path: what "output tokens/s,
LLaMA-7B TP=1, batch 256" looks like when the GEMMs (torch /
hipBLASLt), sampler (argmax) and embedding run too.  Random weights of the
architecture's shapes, batch 256, every sequence at context 1024, caches of
all layers resident.  The operators between the GEMMs are this package's:
rms_norm, rotary_embedding_neox, reshape_and_cache,
single_query_cached_kv_attention, silu_and_mul.

Prints the step time (HIP-graph replay and eager), tokens/s, and the share of
the step spent in each operator class (from a torch profiler pass).

    python tools/bench_decoder_step.py [--model 7b|13b] [--batch 256] [--ctx 1024] [--steps 10]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__  # noqa: E402

MODELS = {"7b": dict(hidden=4096, inter=11008, heads=32, layers=32, vocab=32000),
          "13b": dict(hidden=5120, inter=13824, heads=40, layers=40, vocab=32000)}


def decoder_step(pkg, model="7b", batch=256, ctx=1024, steps=10, num_layers=0, two_call_rotary=False, profile=False,
                 caches=None, block_tables=None, graph=True):
    """Builds the step and times it.  caches: [(key_cache, value_cache)] per layer to run on (fp16, block 16,
    at least batch * ceil(ctx / 16) blocks each) with their block_tables [batch, ceil(ctx / 16)]; default: allocated
    here.  Returns a dict: eager_ms, graph_ms, tokens_per_s, the bytes a step must stream, and (profile=True) the
    device time per operator class."""
    from vllm_amd.input_metadata import InputMetadata
    from vllm_amd.layers import PagedAttentionWithRoPE, RMSNorm, SiluAndMul

    m = MODELS[model]
    hidden, inter, H, vocab = m["hidden"], m["inter"], m["heads"], m["vocab"]
    L = num_layers or m["layers"]
    if caches is not None:
        L = min(L, len(caches))
    D, B, S, CTX = hidden // H, 16, batch, ctx
    dev, dt = torch.device("cuda", torch.cuda.current_device()), torch.float16
    torch.manual_seed(0)

    def w(*shape):
        return torch.empty(*shape, dtype=dt, device=dev).normal_(0, 0.02)

    embed, lm_head = w(vocab, hidden), w(hidden, vocab)
    layers = []
    bps = (CTX + B - 1) // B
    NB = S * bps
    for l in range(L):
        if caches is not None:
            kc, vc = caches[l]
        else:
            kc = torch.empty(NB, H, D // 8, B, 8, dtype=dt, device=dev).normal_(0, 0.5)
            vc = torch.empty(NB, H, D, B, dtype=dt, device=dev).normal_(0, 0.5)
        layers.append(dict(
            ln1=RMSNorm(hidden).to(device=dev, dtype=dt), ln2=RMSNorm(hidden).to(device=dev, dtype=dt),
            qkv=w(hidden, 3 * hidden), o=w(hidden, hidden), gate_up=w(hidden, 2 * inter), down=w(inter, hidden), kc=kc, vc=vc))
    final_norm = RMSNorm(hidden).to(device=dev, dtype=dt)
    cache_ops = None
    if two_call_rotary:
        import types
        cache_ops = types.SimpleNamespace(reshape_and_cache=pkg.cache_ops.reshape_and_cache)
    attn = PagedAttentionWithRoPE(H, D, D ** -0.5, rotary_dim=D, max_position=2048, cache_ops=cache_ops).to(device=dev, dtype=dt)
    act = SiluAndMul()
    if block_tables is None:
        g = torch.Generator(device="cpu").manual_seed(0)
        block_tables = torch.randperm(NB, generator=g).to(torch.int32).reshape(S, bps).to(dev)
    context_lens = torch.full((S,), CTX, dtype=torch.int32, device=dev)
    pos = CTX - 1
    slot_mapping = (block_tables[:, pos // B].long() * B + pos % B).to(torch.int32)
    positions = torch.full((S,), pos, dtype=torch.long, device=dev)
    meta = InputMetadata([], {}, [], slot_mapping, context_lens, CTX, block_tables)
    tokens = torch.randint(0, vocab, (S,), device=dev)
    next_tokens = torch.empty_like(tokens)

    def step():
        x = embed[tokens]
        for ly in layers:
            h = ly["ln1"](x)
            q, k, v = (h @ ly["qkv"]).chunk(3, dim=-1)
            a = attn(positions, q, k, v, ly["kc"], ly["vc"], meta, None)
            x = x + a @ ly["o"]
            h = ly["ln2"](x)
            x = x + act(h @ ly["gate_up"]) @ ly["down"]
        logits = final_norm(x) @ lm_head
        next_tokens.copy_(logits.argmax(dim=-1))

    for _ in range(2):
        step()
    torch.cuda.synchronize()

    def time_it(fn, n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    res = {"model": f"LLaMA-{model}", "batch": S, "context_len": CTX, "layers": L, "dtype": "f16",
           "rotary_cache_write": "two calls" if two_call_rotary else "one pass"}
    res["eager_ms"] = time_it(step, steps)
    if graph:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            step()
        cg.replay()
        torch.cuda.synchronize()
        res["graph_ms"] = time_it(cg.replay, steps)
    best = min(res["eager_ms"], res.get("graph_ms", res["eager_ms"]))
    res["tokens_per_s"] = S / best * 1e3
    wbytes = sum(t.numel() * 2 for ly in layers for t in (ly["qkv"], ly["o"], ly["gate_up"], ly["down"])) + lm_head.numel() * 2
    kvbytes = L * 2 * S * CTX * H * D * 2
    res["weight_bytes"], res["kv_bytes"] = wbytes, kvbytes
    res["ms_at_8_TBps"] = (wbytes + kvbytes) / 8e12 * 1e3
    if profile:
        from torch.profiler import ProfilerActivity, profile as torch_profile
        with torch_profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
        groups = {}
        for ev in prof.key_averages():
            name = ev.key
            t = getattr(ev, "device_time_total", None) or getattr(ev, "cuda_time_total", 0)
            if "pa_decode" in name:
                key = "paged attention (decode)"
            elif "pa_rotary_append" in name:
                key = "rotary + reshape_and_cache (one pass)"
            elif "pa_token" in name:
                key = "reshape_and_cache"
            elif "rms_norm" in name or "rotary" in name or "silu" in name:
                key = "rms_norm / rotary / silu_and_mul"
            elif "Cijk" in name or "gemm" in name.lower() or "hipblas" in name.lower():
                key = "GEMMs (hipBLASLt / rocBLAS)"
            else:
                key = "other (adds, embedding, argmax, copies)"
            groups[key] = groups.get(key, 0.0) + t / 3
        res["device_ms_by_class"] = {k: v / 1e3 for k, v in groups.items()}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--two-call-rotary", action="store_true",
                    help="rotary_embedding_neox then reshape_and_cache as two calls (the reference's sequence) instead "
                         "of the one-pass operator for the generation tokens")
    args = ap.parse_args()
    pkg = __graft_entry__.load_package()
    r = decoder_step(pkg, args.model, args.batch, args.ctx, args.steps, args.layers, args.two_call_rotary,
                     profile=not args.no_profile)
    print(f"{r['model']} decode step, batch {r['batch']}, ctx {r['context_len']}, {r['layers']} layers, fp16, random weights, "
          f"rotary + reshape_and_cache: {r['rotary_cache_write']}")
    print(f"  eager : {r['eager_ms']:8.2f} ms/step  {r['batch'] / r['eager_ms'] * 1e3:10.0f} tokens/s")
    print(f"  graph : {r['graph_ms']:8.2f} ms/step  {r['batch'] / r['graph_ms'] * 1e3:10.0f} tokens/s")
    print(f"  bytes that must stream per step: weights {r['weight_bytes'] / 1e9:.1f} GB + KV {r['kv_bytes'] / 1e9:.1f} GB "
          f"= {r['ms_at_8_TBps']:.2f} ms at 8 TB/s")
    if "device_ms_by_class" in r:
        total = sum(r["device_ms_by_class"].values())
        print("  device time per step by operator class (torch profiler, eager):")
        for k, v in sorted(r["device_ms_by_class"].items(), key=lambda kv: -kv[1]):
            print(f"    {k:42s} {v:8.2f} ms  {100 * v / total:5.1f} %")


if __name__ == "__main__":
    main()
