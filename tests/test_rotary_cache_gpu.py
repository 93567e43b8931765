"""rotary_embedding_neox + reshape_and_cache in one call
(pa_rotary_embedding_neox_reshape_and_cache): query, key and both caches must be
bit-identical to the oracle's rotary_embedding_neox followed by its
reshape_and_cache — the two operators the reference runs one after the other
(vllm/model_executor/layers/attention.py:200-206, :134-140), each of which is
pinned by reference-generated fixtures of its own."""
import numpy as np
import pytest

from oracle import datagen, pa_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(dtype, T, H, D, B, rot, NB, seed, slots=None):
    es = datagen.ELEM_SIZE[dtype]
    x = 16 // es
    base = seed * 1000 + 300
    qkv = datagen.to_storage(datagen.normal(max(T, 1) * 3 * H * D, base + 1), dtype).reshape(max(T, 1), 3, H, D)[:T]
    kc = datagen.to_storage(datagen.normal(NB * H * D * B, base + 2), dtype).reshape(NB, H, D // x, B, x)
    vc = datagen.to_storage(datagen.normal(NB * H * D * B, base + 3), dtype).reshape(NB, H, D, B)
    pos = datagen.randint(max(T, 1), base + 4, 0, 199)[:T]
    cs = datagen.cos_sin_cache(200, rot, dtype)
    if slots is None:  # one token per block, as in a decode step
        blocks = datagen.permutation(NB, base + 5)[:T]
        slots = (blocks * B + datagen.randint(max(T, 1), base + 6, 0, B - 1)[:T]).astype(np.int32)
    return qkv, kc, vc, pos.astype(np.int64), cs, np.asarray(slots, dtype=np.int32)


def _check(pkg, dtype, T, H, D, B, rot, NB, seed, slots=None, via="capi"):
    import torch
    qkv, kc, vc, pos, cs, slots = _inputs(dtype, T, H, D, B, rot, NB, seed, slots)
    # oracle: rotate q and k in place, then scatter k and v
    want = qkv.copy()
    q2, k2 = want[:, 0].reshape(T, H * D), want[:, 1].reshape(T, H * D)  # views of `want`
    pa_oracle.rotary_embedding_neox(pos, q2, k2, D, cs, dtype)
    kc_w, vc_w = kc.copy(), vc.copy()
    keep = slots >= 0  # the oracle has no "no slot" case: skip those tokens by hand
    if keep.any():
        idx = np.nonzero(keep)[0]
        sub = np.ascontiguousarray(want[idx])
        pa_oracle.reshape_and_cache(sub[:, 1], sub[:, 2], kc_w, vc_w, slots[idx], dtype)
    t = datagen.storage_to_torch
    qkv_d = t(qkv.copy(), dtype).to(DEV)
    kc_d, vc_d = t(kc.copy(), dtype).to(DEV), t(vc.copy(), dtype).to(DEV)
    pos_d, cs_d, sl_d = torch.from_numpy(pos).to(DEV), t(cs, dtype).to(DEV), torch.from_numpy(slots).to(DEV)
    fn = (pkg._capi.rotary_embedding_neox_reshape_and_cache if via == "capi"
          else pkg.cache_ops.rotary_embedding_neox_reshape_and_cache)
    fn(pos_d, qkv_d[:, 0], qkv_d[:, 1], qkv_d[:, 2], kc_d, vc_d, sl_d, cs_d)
    torch.cuda.synchronize()
    what = f"{dtype} T={T} H={H} D={D} B={B} rot={rot}"
    assert np.array_equal(datagen.torch_to_storage(qkv_d, dtype), want), f"{what}: query / key rows differ"
    assert np.array_equal(datagen.torch_to_storage(kc_d, dtype), kc_w), f"{what}: key cache differs"
    assert np.array_equal(datagen.torch_to_storage(vc_d, dtype), vc_w), f"{what}: value cache differs"


@pytest.mark.parametrize("dtype", datagen.DTYPES)
@pytest.mark.parametrize("shape", [
    # (T, H, D, B, rot): decode-sized batches take the one-pass kernel ...
    (37, 5, 128, 16, 128), (256, 8, 128, 16, 128), (16, 3, 64, 8, 64), (100, 4, 96, 32, 96), (64, 2, 80, 16, 80),
    (48, 16, 96, 16, 24),   # GPT-NeoX: rotary_pct 0.25 (vllm/model_executor/models/gpt_neox.py:64)
    (40, 3, 128, 16, 32),
    # ... everything else the two operators: a handful of tokens, a prompt-sized batch, a rot_dim
    # whose half is not a whole 16-byte vector
    (1, 4, 128, 16, 128), (15, 4, 64, 16, 64), (600, 2, 64, 16, 64), (33, 2, 64, 16, 20), (0, 2, 64, 16, 64),
], ids=lambda s: "T%d_H%d_D%d_B%d_rot%d" % s)
def test_rotary_reshape_and_cache_matches_the_two_oracle_steps(pkg, dtype, shape):
    T, H, D, B, rot = shape
    if dtype == "float32" and rot == 20:
        pytest.skip("covered by the 16-bit types (fp32 vectors are 4 elements: 10 is not a multiple either way)")
    _check(pkg, dtype, T, H, D, B, rot, NB=max(T, 1) + 3, seed=T + H + rot)


def test_rotary_reshape_and_cache_tokens_sharing_a_block_and_tokens_without_a_slot(pkg):
    """Two tokens in one block (a prompt's run) make the value side fall back to
    element stores for that block; a negative slot rotates the token and caches
    nothing; via the cache_ops module."""
    T, H, D, B = 40, 4, 128, 16
    slots = [b * B + (b % B) for b in range(T)]
    slots[5] = slots[4] + 1        # same block as token 4
    slots[20] = -1
    slots[39] = slots[38] - 3 if slots[38] % B >= 3 else slots[38] + 1
    for dtype in ("float16", "bfloat16", "float32"):
        _check(pkg, dtype, T, H, D, B, 128, NB=T + 2, seed=9, slots=slots, via="module")


def test_rope_layer_with_the_one_pass_cache_write_equals_the_two_call_layer(pkg):
    """PagedAttentionWithRoPE on a mixed batch (prompts, then generation tokens,
    then padding): with the one-pass operator for the generation tokens the
    layer's output and caches are bit-identical to the same layer restricted to
    the reference's two calls."""
    import types
    import torch
    from vllm_amd.layers import PagedAttentionWithRoPE
    from vllm_amd.worker import prepare_inputs
    from tests.test_model_gpu import SeqData
    from types import SimpleNamespace
    H, D, B, NB = 4, 64, 16, 64
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float16)
    try:
        fused_layer = PagedAttentionWithRoPE(H, D, D ** -0.5, rotary_dim=D, max_position=256).to(DEV)
        two_call_ops = types.SimpleNamespace(reshape_and_cache=pkg.cache_ops.reshape_and_cache)
        plain_layer = PagedAttentionWithRoPE(H, D, D ** -0.5, rotary_dim=D, max_position=256,
                                             cache_ops=two_call_ops).to(DEV)
    finally:
        torch.set_default_dtype(old)
    g = torch.Generator(device=DEV)
    g.manual_seed(4)
    groups, block = [], 1
    for i, n in enumerate([5, 33]):  # prompts
        nb = (n + B - 1) // B
        groups.append(SimpleNamespace(request_id=str(i), is_prompt=True, seq_data={i: SeqData(list(range(n)))},
                                      sampling_params=None, block_tables={i: list(range(block, block + nb))}))
        block += nb
    for i in range(2, 22):  # 20 generation tokens: the one-pass kernel's range
        n = 3 + 7 * i
        nb = (n + B - 1) // B
        groups.append(SimpleNamespace(request_id=str(i), is_prompt=False, seq_data={i: SeqData(list(range(n)))},
                                      sampling_params=None, block_tables={i: list(range(block, block + nb))}))
        block += nb
    assert block <= 4 * NB
    tokens, positions, meta = prepare_inputs(groups, B, device=DEV)
    T = tokens.shape[0]
    qkv = torch.randn(T, 3 * H * D, dtype=torch.float16, device=DEV, generator=g)
    results = []
    for layer in (fused_layer, plain_layer):
        kc = torch.randn(4 * NB, H, D // 8, B, 8, dtype=torch.float16, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
        vc = torch.randn(4 * NB, H, D, B, dtype=torch.float16, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
        x = qkv.clone()
        q, k, v = x.chunk(3, dim=-1)
        _, _, m = prepare_inputs(groups, B, device=DEV)
        out = layer(positions, q, k, v, kc, vc, m, None)
        torch.cuda.synchronize()
        results.append((out[:meta.num_valid_tokens].clone(), kc, vc))
    for a, b in zip(results[0], results[1]):
        assert torch.equal(a, b)
