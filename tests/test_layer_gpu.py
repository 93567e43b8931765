"""GPU tests of the host-side callers on top of the HIP modules:
PagedAttention.forward over several engine steps, CacheEngine swap / copy."""
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


def _dense_attention(q, K, V, scale):
    """q [H, D], K/V [n, H, D] (fp32) -> [H, D]: softmax(scale q K^T) V."""
    import torch
    logits = torch.einsum("hd,nhd->hn", q, K) * scale
    p = torch.softmax(logits, dim=-1)
    return torch.einsum("hn,nhd->hd", p, V)


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
def test_paged_attention_prompt_then_decode_steps_match_dense_attention(pkg, dtype_name):
    """Three sequences: prompt step, then 40 decode steps (crossing several
    block boundaries), one layer of an engine with two layers of cache.  Every
    output row is checked against dense attention over the sequence's full
    K/V history."""
    import torch
    from vllm_amd.layers import PagedAttention
    from vllm_amd.worker import CacheEngine, prepare_inputs, simple_configs

    dtype = getattr(torch, dtype_name)
    H, D, B, L = 4, 64, 16, 2
    scale = D ** -0.5
    cc, mc, pc = simple_configs(L, H, D, dtype, B, num_gpu_blocks=64, num_cpu_blocks=4)
    engine = CacheEngine(cc, mc, pc)
    layer = PagedAttention(H, D, scale)
    kc, vc = engine.gpu_cache[1]
    kc.fill_(float("nan"))  # an uninitialised cache may hold anything
    vc.fill_(float("nan"))
    g = torch.Generator(device=DEV)
    g.manual_seed(0)

    prompt_lens = [5, 17, 33]
    seqs = {i: SeqData(range(100 * i, 100 * i + n)) for i, n in enumerate(prompt_lens)}
    free_blocks = list(range(63, 0, -1))  # hand out blocks in a scattered order
    tables = {i: [free_blocks.pop() for _ in range((n + B - 1) // B)] for i, n in enumerate(prompt_lens)}
    hist_k = {i: [] for i in seqs}
    hist_v = {i: [] for i in seqs}
    rtol = 3e-3 if dtype_name == "float16" else 2e-2

    def run_step(groups, rows_of):
        tokens, positions, meta = prepare_inputs(groups, B, device=DEV)
        n_pad = tokens.shape[0]
        qkv = torch.randn(n_pad, 3 * H * D, dtype=dtype, device=DEV, generator=g) * 0.5
        q, k, v = qkv.chunk(3, dim=-1)  # column slices, like llama.py:119
        out = layer(q, k, v, kc, vc, meta, None)
        torch.cuda.synchronize()
        assert out.shape == (n_pad, H * D)
        q3, k3, v3, o3 = (t.reshape(n_pad, H, D).float() for t in (q, k, v, out))
        for row, sid in rows_of:
            hist_k[sid].append(k3[row])
            hist_v[sid].append(v3[row])
        return q3, o3

    # prompt step: three prompt groups
    groups = [SimpleNamespace(request_id=str(i), is_prompt=True, seq_data={i: seqs[i]},
                              sampling_params=None, block_tables={i: tables[i]}) for i in seqs]
    rows, r = [], 0
    for i, n in enumerate(prompt_lens):
        rows += [(r + t, i) for t in range(n)]
        r += n
    q3, o3 = run_step(groups, rows)
    r = 0
    for i, n in enumerate(prompt_lens):
        K, V = torch.stack(hist_k[i]), torch.stack(hist_v[i])
        for t in range(n):
            want = _dense_attention(q3[r + t], K[:t + 1], V[:t + 1], scale)
            assert torch.allclose(o3[r + t], want, atol=2e-3, rtol=rtol), (i, t)
        r += n

    # decode steps
    for step in range(40):
        for i in seqs:
            seqs[i].tokens.append(1000 + step)
            if len(seqs[i].tokens) > len(tables[i]) * B:
                tables[i].append(free_blocks.pop())
        groups = [SimpleNamespace(request_id=str(i), is_prompt=False, seq_data={i: seqs[i]},
                                  sampling_params=None, block_tables={i: tables[i]}) for i in seqs]
        q3, o3 = run_step(groups, [(row, i) for row, i in enumerate(seqs)])
        for row, i in enumerate(seqs):
            want = _dense_attention(q3[row], torch.stack(hist_k[i]), torch.stack(hist_v[i]), scale)
            assert torch.allclose(o3[row], want, atol=2e-3, rtol=rtol), (step, i)
    # the other layer's cache was never touched
    assert engine.gpu_cache[0][0].data_ptr() != kc.data_ptr()


def test_cache_engine_swap_out_swap_in_and_copy(pkg):
    import torch
    from vllm_amd.worker import CacheEngine, simple_configs

    L, H, D, B = 3, 2, 64, 16
    cc, mc, pc = simple_configs(L, H, D, torch.float16, B, num_gpu_blocks=10, num_cpu_blocks=5)
    engine = CacheEngine(cc, mc, pc)
    assert engine.get_key_block_shape() == (H, D // 8, B, 8)
    assert engine.get_value_block_shape() == (H, D, B)
    assert len(engine.gpu_cache) == L and len(engine.events) == L
    assert engine.gpu_cache[0][0].shape == (10, H, D // 8, B, 8) and engine.cpu_cache[0][1].shape == (5, H, D, B)
    assert engine.cpu_cache[0][0].is_pinned()
    g = torch.Generator(device=DEV)
    g.manual_seed(1)
    for k, v in engine.gpu_cache:
        k.copy_(torch.randn(k.shape, dtype=k.dtype, device=DEV, generator=g))
        v.copy_(torch.randn(v.shape, dtype=v.dtype, device=DEV, generator=g))
    before = [(k.clone(), v.clone()) for k, v in engine.gpu_cache]
    torch.cuda.synchronize()

    engine.swap_out({2: 0, 3: 1, 9: 4})
    for e in engine.events:
        e.synchronize()
    for l in range(L):
        for s, d in {2: 0, 3: 1, 9: 4}.items():
            assert torch.equal(engine.cpu_cache[l][0][d], before[l][0][s].cpu())
            assert torch.equal(engine.cpu_cache[l][1][d], before[l][1][s].cpu())
    for k, v in engine.gpu_cache:
        k[[2, 3, 9]] = 0
        v[[2, 3, 9]] = 0
    torch.cuda.synchronize()
    engine.swap_in({0: 2, 1: 3, 4: 9})
    for e in engine.events:
        e.wait()  # what the attention layer does (attention.py:125-126)
    torch.cuda.synchronize()
    for l in range(L):
        assert torch.equal(engine.gpu_cache[l][0], before[l][0])
        assert torch.equal(engine.gpu_cache[l][1], before[l][1])

    engine.copy({1: [5, 6], 4: [7]})
    torch.cuda.synchronize()
    for l in range(L):
        for s, ds in {1: [5, 6], 4: [7]}.items():
            for d in ds:
                assert torch.equal(engine.gpu_cache[l][0][d], before[l][0][s])
                assert torch.equal(engine.gpu_cache[l][1][d], before[l][1][s])
        assert torch.equal(engine.gpu_cache[l][0][8], before[l][0][8])


def test_rope_attention_rmsnorm_swiglu_layers_over_decode_steps(pkg):
    """A LLaMA-shaped slice of one decoder layer (llama.py:88-131, 56-81):
    RMSNorm -> qkv -> PagedAttentionWithRoPE -> gate/up -> SiluAndMul, one
    prompt step and 20 decode steps for two sequences, against the same
    arithmetic in fp32 torch with dense attention over the rotated history."""
    import torch
    from vllm_amd.layers import PagedAttentionWithRoPE, RMSNorm, SiluAndMul
    from vllm_amd.worker import prepare_inputs

    dtype = torch.float16
    H, D, B, hidden, inter = 4, 128, 16, 512, 768
    scale = D ** -0.5
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)  # the cos/sin table takes the default dtype (attention.py:181-184)
    try:
        attn = PagedAttentionWithRoPE(H, D, scale, rotary_dim=D, max_position=256).to(DEV)
    finally:
        torch.set_default_dtype(old)
    assert attn.cos_sin_cache.dtype == dtype and attn.cos_sin_cache.shape == (256, D)
    norm = RMSNorm(hidden).to(DEV).to(dtype)
    norm.weight.data.copy_(1 + 0.1 * torch.randn(hidden, device=DEV, generator=g))
    act = SiluAndMul()
    w_qkv = (torch.randn(hidden, 3 * H * D, device=DEV, generator=g) * hidden ** -0.5).to(dtype)
    w_gu = (torch.randn(H * D, 2 * inter, device=DEV, generator=g) * (H * D) ** -0.5).to(dtype)
    kc = torch.full((32, H, D // 8, B, 8), float("nan"), dtype=dtype, device=DEV)
    vc = torch.full((32, H, D, B), float("nan"), dtype=dtype, device=DEV)

    def rotate(x, pos):  # x [n, H, D] fp32, NeoX pairing (i, i + D/2)
        cs = attn.cos_sin_cache[pos].float()
        c, s = cs[:, None, : D // 2], cs[:, None, D // 2:]
        x1, x2 = x[..., : D // 2], x[..., D // 2:]
        return torch.cat((x1 * c - x2 * s, x2 * c + x1 * s), dim=-1)

    seqs = {0: SeqData(range(7)), 1: SeqData(range(50, 50 + 21))}
    tables = {0: [9, 3], 1: [17, 4, 11]}
    hist = {0: ([], []), 1: ([], [])}

    def step(is_prompt):
        groups = [SimpleNamespace(request_id=str(i), is_prompt=is_prompt, seq_data={i: seqs[i]},
                                  sampling_params=None, block_tables={i: tables[i]}) for i in seqs]
        _, positions, meta = prepare_inputs(groups, B, device=DEV)
        n = positions.shape[0]
        x = torch.randn(n, hidden, dtype=dtype, device=DEV, generator=g)
        h = norm(x)
        h_ref = (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6)).to(dtype) * norm.weight.data
        assert torch.allclose(h, h_ref, atol=1e-3, rtol=1e-3)
        qkv = h @ w_qkv
        q, k, v = qkv.chunk(3, dim=-1)
        q0, k0 = q.clone().float().view(n, H, D), k.clone().float().view(n, H, D)
        o = attn(positions, q, k, v, kc, vc, meta, None)
        y = act(o @ w_gu)
        torch.cuda.synchronize()
        qr, kr = rotate(q0, positions), rotate(k0, positions)
        assert torch.allclose(q.float().view(n, H, D), qr, atol=2e-3, rtol=2e-3)  # rotated in place
        rows = []
        if is_prompt:
            r = 0
            for i in seqs:
                for t in range(seqs[i].get_len()):
                    rows.append((r, i))
                    r += 1
        else:
            rows = [(r, i) for r, i in enumerate(seqs)]
        v3 = v.float().view(n, H, D)
        for r, i in rows:
            hist[i][0].append(kr[r])
            hist[i][1].append(v3[r])
            upto = positions[r].item() + 1
            want = _dense_attention(qr[r], torch.stack(hist[i][0][:upto]), torch.stack(hist[i][1][:upto]), scale)
            assert torch.allclose(o.float().view(n, H, D)[r], want, atol=3e-3, rtol=3e-3), (is_prompt, r, i)
        valid = [r for r, _ in rows]
        gu = (o @ w_gu).float()[valid]
        want_y = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]
        assert torch.allclose(y.float()[valid], want_y, atol=2e-3, rtol=2e-3)

    step(True)
    for s in range(20):
        for i in seqs:
            seqs[i].tokens.append(900 + s)
            if seqs[i].get_len() > len(tables[i]) * B:
                tables[i].append(20 + i * 5 + len(tables[i]))
        step(False)
