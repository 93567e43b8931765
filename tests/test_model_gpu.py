"""System test: a small LLaMA-shaped decoder (4 layers, hidden 256, 4 heads x
64, SwiGLU, RMSNorm, rotary) generating greedily through the paged path —
prompt step with prompt attention + cache writes, then decode steps with
paged attention, block tables handed out in scattered order, one sequence
forked mid-way (copy-on-write through copy_blocks) — against the same
weights evaluated densely in fp32 torch with no cache at all (every step
recomputes the whole sequence).  Checks logits at every step and that the
greedy tokens agree wherever the reference's top-2 margin is not a tie."""
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class SeqData:
    def __init__(self, tokens):
        self.tokens = list(tokens)

    def get_len(self):
        return len(self.tokens)

    def get_token_ids(self):
        return self.tokens

    def get_last_token_id(self):
        return self.tokens[-1]


def test_small_llama_greedy_generation_matches_dense_recompute(pkg):
    import torch
    from vllm_amd.layers import PagedAttentionWithRoPE, RMSNorm, SiluAndMul
    from vllm_amd.worker import CacheEngine, prepare_inputs, simple_configs

    dtype = torch.float16
    L, hidden, H, D, inter, vocab, B = 4, 256, 4, 64, 512, 500, 16
    g = torch.Generator(device=DEV)
    g.manual_seed(2)

    def w(*shape, std):
        return (torch.randn(*shape, device=DEV, generator=g) * std).to(dtype)

    embed = w(vocab, hidden, std=1.0)
    lm_head = w(hidden, vocab, std=hidden ** -0.5)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        attn = PagedAttentionWithRoPE(H, D, D ** -0.5, rotary_dim=D, max_position=128).to(DEV)
    finally:
        torch.set_default_dtype(old)
    act = SiluAndMul()
    layers = []
    for _ in range(L):
        ln1, ln2 = RMSNorm(hidden).to(DEV).to(dtype), RMSNorm(hidden).to(DEV).to(dtype)
        ln1.weight.data.copy_(1 + 0.1 * torch.randn(hidden, device=DEV, generator=g))
        ln2.weight.data.copy_(1 + 0.1 * torch.randn(hidden, device=DEV, generator=g))
        layers.append(dict(ln1=ln1, ln2=ln2, qkv=w(hidden, 3 * hidden, std=hidden ** -0.5), o=w(hidden, hidden, std=hidden ** -0.5),
                           gate_up=w(hidden, 2 * inter, std=hidden ** -0.5), down=w(inter, hidden, std=inter ** -0.5)))
    final_norm = RMSNorm(hidden).to(DEV).to(dtype)
    cc, mc, pc = simple_configs(L, H, D, dtype, B, num_gpu_blocks=48, num_cpu_blocks=2)
    engine = CacheEngine(cc, mc, pc)
    for kc, vc in engine.gpu_cache:
        kc.fill_(float("nan"))
        vc.fill_(float("nan"))

    def paged_forward(tokens, positions, meta):
        x = embed[tokens]
        for i, ly in enumerate(layers):
            kc, vc = engine.gpu_cache[i]
            q, k, v = (ly["ln1"](x) @ ly["qkv"]).chunk(3, dim=-1)
            x = x + attn(positions, q, k, v, kc, vc, meta, None) @ ly["o"]
            x = x + act(ly["ln2"](x) @ ly["gate_up"]) @ ly["down"]
        return final_norm(x) @ lm_head

    def dense_logits(token_ids):
        """fp32, no cache: logits of the last position of one sequence."""
        t = torch.tensor(token_ids, device=DEV)
        n = len(token_ids)
        x = embed[t].float()
        cs = attn.cos_sin_cache[:n].float()
        c, s = cs[:, None, : D // 2], cs[:, None, D // 2:]

        def rms(v, wt):
            return v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-6) * wt.float()

        def rot(u):
            u1, u2 = u[..., : D // 2], u[..., D // 2:]
            return torch.cat((u1 * c - u2 * s, u2 * c + u1 * s), dim=-1)

        for ly in layers:
            q, k, v = (rms(x, ly["ln1"].weight.data) @ ly["qkv"].float()).chunk(3, dim=-1)
            q, k, v = rot(q.view(n, H, D)), rot(k.view(n, H, D)), v.view(n, H, D)
            sc = torch.einsum("qhd,khd->hqk", q, k) * D ** -0.5
            sc = sc + torch.full((n, n), float("-inf"), device=DEV).triu(1)
            a = torch.einsum("hqk,khd->qhd", torch.softmax(sc, dim=-1), v).reshape(n, hidden)
            x = x + a @ ly["o"].float()
            gu = rms(x, ly["ln2"].weight.data) @ ly["gate_up"].float()
            x = x + (torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]) @ ly["down"].float()
        return rms(x[-1], final_norm.weight.data) @ lm_head.float()

    free_blocks = list(range(47, 0, -1))
    seqs = {0: SeqData([3, 77, 12, 400, 9]), 1: SeqData(list(range(100, 137))), 2: SeqData([250] * 17)}
    tables = {i: [free_blocks.pop() for _ in range((s.get_len() + B - 1) // B)] for i, s in seqs.items()}

    def check_and_extend(logits_rows, ids):
        disagreements = 0
        for row, i in zip(logits_rows, ids):
            want = dense_logits(seqs[i].tokens)
            assert torch.allclose(row.float(), want, atol=3e-2, rtol=3e-2), (i, seqs[i].get_len(),
                                                                             float((row.float() - want).abs().max()))
            top2 = want.topk(2).values
            tok_ref, tok = int(want.argmax()), int(row.argmax())
            if float(top2[0] - top2[1]) > 0.1:
                assert tok == tok_ref, (i, seqs[i].get_len())
            else:
                disagreements += tok != tok_ref
            seqs[i].tokens.append(tok_ref)  # follow the reference so both paths see the same history
        return disagreements

    # prompt step
    ids = list(seqs)
    groups = [SimpleNamespace(request_id=str(i), is_prompt=True, seq_data={i: seqs[i]}, sampling_params=None,
                              block_tables={i: tables[i]}) for i in ids]
    tokens, positions, meta = prepare_inputs(groups, B, device=DEV)
    logits = paged_forward(tokens, positions, meta)
    ends, r = [], 0
    for i in ids:
        r += seqs[i].get_len()
        ends.append(r - 1)
    check_and_extend([logits[e] for e in ends], ids)

    # decode steps; after 6 of them sequence 1 is forked into sequence 3 (its last,
    # partially filled block is copied: copy-on-write as in block_manager.py:126-135)
    for step in range(24):
        if step == 6:
            seqs[3] = SeqData(seqs[1].tokens)
            shared, last = tables[1][:-1], tables[1][-1]
            new_block = free_blocks.pop()
            engine.copy({last: [new_block]})
            tables[3] = shared + [new_block]
        ids = list(seqs)
        for i in ids:
            if seqs[i].get_len() > len(tables[i]) * B:
                tables[i].append(free_blocks.pop())
        groups = [SimpleNamespace(request_id=str(i), is_prompt=False, seq_data={i: seqs[i]}, sampling_params=None,
                                  block_tables={i: tables[i]}) for i in ids]
        tokens, positions, meta = prepare_inputs(groups, B, device=DEV)
        logits = paged_forward(tokens, positions, meta)
        check_and_extend([logits[r] for r in range(len(ids))], ids)
        if step == 6:  # the fork sees exactly what its parent sees
            assert torch.equal(logits[ids.index(1)], logits[ids.index(3)])


def test_opt125m_shaped_greedy_generation_paged_vs_dense(pkg):
    """OPT-125m's attention shape — 12 heads x 64, hidden 768, learned position
    embeddings (offset 2), pre-LayerNorm, biased projections, ReLU FFN 3072, tied
    lm_head; plain PagedAttention without rotary
    (vllm/model_executor/models/opt.py:38-91, 94-149, 211-229) — two layers with
    random weights.  Greedy token ids through the paged path (prompt step, then
    16 decode steps on scattered block tables) against the same weights
    evaluated densely in fp32 with no cache."""
    import torch
    import torch.nn.functional as F
    from vllm_amd.layers import PagedAttention
    from vllm_amd.worker import CacheEngine, prepare_inputs, simple_configs

    dtype = torch.float16
    L, hidden, H, D, ffn, vocab, B, max_pos = 2, 768, 12, 64, 3072, 1000, 16, 128
    g = torch.Generator(device=DEV)
    g.manual_seed(5)

    def w(*shape, std):
        return (torch.randn(*shape, device=DEV, generator=g) * std).to(dtype)

    embed = w(vocab, hidden, std=0.1)  # tied lm_head: logits of a layer-normed row then have a spread of about 3
    pos_embed = w(max_pos + 2, hidden, std=0.03)  # OPTLearnedPositionalEmbedding: rows are position + 2
    attn = PagedAttention(H, D, D ** -0.5)
    layers = []
    for _ in range(L):
        layers.append(dict(
            ln1_w=1 + 0.1 * torch.randn(hidden, device=DEV, generator=g), ln1_b=0.1 * torch.randn(hidden, device=DEV, generator=g),
            ln2_w=1 + 0.1 * torch.randn(hidden, device=DEV, generator=g), ln2_b=0.1 * torch.randn(hidden, device=DEV, generator=g),
            qkv=w(hidden, 3 * hidden, std=hidden ** -0.5), qkv_b=w(3 * hidden, std=0.1),
            o=w(hidden, hidden, std=hidden ** -0.5), o_b=w(hidden, std=0.1),
            fc1=w(hidden, ffn, std=hidden ** -0.5), fc1_b=w(ffn, std=0.1),
            fc2=w(ffn, hidden, std=ffn ** -0.5), fc2_b=w(hidden, std=0.1)))
    fin_w = 1 + 0.1 * torch.randn(hidden, device=DEV, generator=g)
    fin_b = 0.1 * torch.randn(hidden, device=DEV, generator=g)
    cc, mc, pc = simple_configs(L, H, D, dtype, B, num_gpu_blocks=40, num_cpu_blocks=2)
    engine = CacheEngine(cc, mc, pc)
    for kc, vc in engine.gpu_cache:
        kc.fill_(float("nan"))
        vc.fill_(float("nan"))

    def ln(x, wt, b):
        return F.layer_norm(x, (hidden,), wt.to(x.dtype), b.to(x.dtype))

    def paged_forward(tokens, positions, meta):
        x = embed[tokens] + pos_embed[positions + 2]
        for i, ly in enumerate(layers):
            kc, vc = engine.gpu_cache[i]
            q, k, v = (ln(x, ly["ln1_w"], ly["ln1_b"]) @ ly["qkv"] + ly["qkv_b"]).chunk(3, dim=-1)
            x = x + attn(q, k, v, kc, vc, meta, None) @ ly["o"] + ly["o_b"]
            x = x + torch.relu(ln(x, ly["ln2_w"], ly["ln2_b"]) @ ly["fc1"] + ly["fc1_b"]) @ ly["fc2"] + ly["fc2_b"]
        return ln(x, fin_w, fin_b) @ embed.t()

    def dense_logits(token_ids):
        n = len(token_ids)
        t = torch.tensor(token_ids, device=DEV)
        x = embed[t].float() + pos_embed[torch.arange(n, device=DEV) + 2].float()
        for ly in layers:
            q, k, v = (ln(x, ly["ln1_w"], ly["ln1_b"]) @ ly["qkv"].float() + ly["qkv_b"].float()).chunk(3, dim=-1)
            q, k, v = q.view(n, H, D), k.view(n, H, D), v.view(n, H, D)
            sc = torch.einsum("qhd,khd->hqk", q, k) * D ** -0.5
            sc = sc + torch.full((n, n), float("-inf"), device=DEV).triu(1)
            a = torch.einsum("hqk,khd->qhd", torch.softmax(sc, dim=-1), v).reshape(n, hidden)
            x = x + a @ ly["o"].float() + ly["o_b"].float()
            x = x + torch.relu(ln(x, ly["ln2_w"], ly["ln2_b"]) @ ly["fc1"].float() + ly["fc1_b"].float()) @ ly["fc2"].float() \
                + ly["fc2_b"].float()
        return ln(x[-1], fin_w, fin_b) @ embed.float().t()

    free_blocks = list(range(39, 0, -1))
    seqs = {0: SeqData([2, 45, 901, 7]), 1: SeqData(list(range(300, 333))), 2: SeqData([17] * 16), 3: SeqData([5])}
    tables = {i: [free_blocks.pop() for _ in range((s.get_len() + B - 1) // B)] for i, s in seqs.items()}
    paged_ids = {i: [] for i in seqs}
    dense_ids = {i: [] for i in seqs}

    def step(is_prompt):
        ids = list(seqs)
        for i in ids:
            if seqs[i].get_len() > len(tables[i]) * B:
                tables[i].append(free_blocks.pop())
        groups = [SimpleNamespace(request_id=str(i), is_prompt=is_prompt, seq_data={i: seqs[i]}, sampling_params=None,
                                  block_tables={i: tables[i]}) for i in ids]
        tokens, positions, meta = prepare_inputs(groups, B, device=DEV)
        logits = paged_forward(tokens, positions, meta)
        rows, r = [], 0
        for i in ids:
            r += seqs[i].get_len() if is_prompt else 1
            rows.append(r - 1)
        for row, i in zip(rows, ids):
            want = dense_logits(seqs[i].tokens)
            got = logits[row].float()
            assert torch.allclose(got, want, atol=5e-2, rtol=3e-2), (i, seqs[i].get_len(), float((got - want).abs().max()))
            top2 = want.topk(2).values
            tok_ref, tok = int(want.argmax()), int(got.argmax())
            if float(top2[0] - top2[1]) > 0.15:  # not a near-tie at fp16 resolution of the logits
                assert tok == tok_ref, (i, seqs[i].get_len())
                paged_ids[i].append(tok)
                dense_ids[i].append(tok_ref)
            seqs[i].tokens.append(tok_ref)

    step(True)
    for _ in range(16):
        step(False)
    assert paged_ids == dense_ids
    assert sum(len(v) for v in paged_ids.values()) >= 4 * 17 * 0.8, "too many near-ties: the comparison is vacuous"


def test_throughput_harness_serves_every_request_on_a_tiny_model(pkg):
    """tools/bench_throughput.py (the reference's offline-throughput benchmark
    restated): continuous batching of mixed prompt + generation steps through
    the paged path.  On a tiny model with a tight token budget and few cache
    blocks: every request gets exactly its output length, and all blocks come
    back."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench_throughput.py")
    spec = importlib.util.spec_from_file_location("bench_throughput", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.run(model="tiny", num_prompts=40, seed=3, max_num_batched_tokens=1100, max_num_seqs=12,
                  num_blocks=12 * 128 + 64, verbose=False)
    reqs = mod.sample_requests(40, 3)
    assert res["output_tokens"] == sum(o for _, o in reqs) and res["prompt_tokens"] == sum(p for p, _ in reqs)
    assert res["steps"] >= max(o for _, o in reqs)
    assert 0 < res["steps_with_prompts"] <= 40
    assert res["output_tokens_per_s"] > 0


def _load_throughput_harness():
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bench_throughput.py")
    spec = importlib.util.spec_from_file_location("bench_throughput", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("model", ["tiny", "tiny-neox"])
def test_forked_and_swapped_sequences_equal_their_independent_recompute(pkg, model):
    """Parallel sampling (n = 3) on a cache so small that groups are preempted: a request's three sequences
    share its prompt's blocks, the first append to a shared last block copies it (cache_ops.copy_blocks),
    preempted groups go to pinned host memory and come back (cache_ops.swap_blocks both ways, on the cache
    stream, fenced by the per-layer events).  Every sequence — fixed first token (the j-th most likely after
    the prompt), greedy after — must produce the tokens that the same model produces for the same prefix as a
    request of its own (n = 1, prompt = the request's prompt + that first token) with cache to spare: no fork,
    no copy, no swap.  Steps at which either run's two best logits lie within 2e-4 end the comparison of that
    sequence (float32 model; the two runs reach a position through different kernels)."""
    mod = _load_throughput_harness()
    rng = np.random.default_rng(5)
    vocab = mod.MODELS[model]["vocab"]
    # prompt lengths around block boundaries: 16 = a full last block (no copy-on-write), the others partial
    lens = [5, 16, 23, 31, 32, 40, 47, 9, 18, 27]
    outs = [14, 9, 21, 12, 17, 8, 11, 19, 13, 16]
    requests = [(rng.integers(0, vocab, size=p).tolist(), o) for p, o in zip(lens, outs)]
    over = {"weight_std": 0.25}  # logits with margins: greedy continuations are well defined
    common = dict(model=model, seed=2, max_num_batched_tokens=200, dtype_name="float32", verbose=False,
                  sampling="greedy_distinct", cfg_overrides=over, record_margins=True)
    res = mod.run(requests=requests, n=3, max_num_seqs=12, num_blocks=16, num_cpu_blocks=40, **common)
    assert res["forks"] == 2 * len(requests)
    assert res["copy_on_write_blocks"] >= 2 * (len(requests) - 2) and res["copy_blocks_calls"] >= 1
    assert res["swap_out_groups"] >= 1 and res["swap_in_groups"] == res["swap_out_groups"], res
    assert res["swap_out_blocks"] >= 1 and res["swap_in_blocks"] == res["swap_out_blocks"]
    seqs = res["sequences"]
    assert len(seqs) == 3 * len(requests)
    # the same prefixes as requests of their own
    singles = []
    for req, tokens, _ in seqs:
        p = len(requests[req][0])
        singles.append((tokens[:p + 1], requests[req][1] - 1))
    ref = mod.run(requests=singles, n=1, max_num_seqs=64, num_blocks=512, num_cpu_blocks=0, **common)
    assert ref["swap_out_groups"] == 0 and ref["copy_on_write_blocks"] == 0
    ref_by_req = {r: (t, m) for r, t, m in ref["sequences"]}
    compared = total = 0
    for i, (req, tokens, margins) in enumerate(seqs):
        p = len(requests[req][0])
        want, want_margins = ref_by_req[i]
        assert len(tokens) == p + requests[req][1] and len(want) == len(tokens)
        assert want[:p + 1] == tokens[:p + 1]
        # margins[k] belongs to token p + k of the forked run; the single run's first margin to token p + 1
        for k in range(1, requests[req][1]):
            total += 1
            if margins[k] < 2e-4 or want_margins[k - 1] < 2e-4:
                break
            assert tokens[p + k] == want[p + k], (model, req, i, k)
            compared += 1
    assert compared >= 0.9 * total, f"{compared} of {total} tokens compared: too many near-ties, the test is vacuous"
