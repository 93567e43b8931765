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
