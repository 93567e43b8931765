"""Prompt attention (causal attention inside each prompt of a flat batch): the
C oracle against fixtures produced by the reference's own restatement
(tests/kernels/test_attention.py:14-37, 69-101) on CPU, and the matrix-core
kernel against both on the GPU, through the C ABI and the operator module.

Bars.  Against the reference run in the storage type on its own input
distribution: atol 1e-3, rtol 1e-5 (its criterion, test_attention.py:267).
Against the reference formula in fp32 on unit-scale inputs: the kernel rounds
the softmax weights to the storage type before the P.V product (as the
reference's storage-type arithmetic does), so the bound is one rounding of the
weights plus one of the output: |err| <= eps * (sum_j p_j |v_j| + |out|) with
eps = 2^-11 (fp16), 2^-8 (bf16); fp32 is within 2e-5."""
import json
import os

import numpy as np
import pytest

from oracle import datagen, pa_oracle
from tests import helpers

with open(os.path.join(helpers.GOLDEN_DIR, "prompt_golden.json")) as _f:
    META = json.load(_f)
DATA = np.load(os.path.join(helpers.GOLDEN_DIR, "prompt_golden.npz"), allow_pickle=False)
DEV = "cuda:0"
EPS = {"float32": 2e-5, "float16": 2.0 ** -11, "bfloat16": 2.0 ** -8}


def _case(name):
    return datagen.PromptCase(**META[name]["case"])


def _check_against_fixture(name, case, out_storage, n_valid):
    got = datagen.storage_to_float32(out_storage, case.dtype)[:n_valid].astype(np.float64)
    ref_dtype = datagen.storage_to_float32(DATA[name + "/ref_dtype"], case.dtype).astype(np.float64)
    ref_f32 = DATA[name + "/ref_f32"].astype(np.float64)
    assert got.shape == ref_f32.shape and np.all(np.isfinite(got))
    if case.dist == "uniform1e-3":
        assert np.allclose(got, ref_dtype, atol=1e-3, rtol=1e-5)
        assert np.abs(got - ref_f32).max() <= 2e-5 * 1e-3 + EPS[case.dtype] * 2e-3
    else:
        # values are N(0,1): sum_j p_j |v_j| <= max |v| ~ 4.5; use the measured magnitude of the row
        bound = EPS[case.dtype] * (2.0 + 2.0 * np.abs(ref_f32)) + 2e-6
        err = np.abs(got - ref_f32)
        assert np.all(err <= bound), f"{name}: max err {err.max():.3e}, worst excess {(err - bound).max():.3e}"


@pytest.mark.parametrize("name", sorted(META))
def test_oracle_matches_reference_fixture(name):
    case = _case(name)
    inp = datagen.make_prompt_inputs(case)
    assert datagen.checksum_arrays(inp) == META[name]["inputs_sha256"]
    qkv = inp["qkv"]
    out = pa_oracle.multi_query_kv_attention(qkv[:, 0], qkv[:, 1], qkv[:, 2], inp["cu_seqlens"],
                                             case.head_size ** -0.5, case.dtype)
    n = int(inp["cu_seqlens"][-1])
    _check_against_fixture(name, case, out, n)
    assert not out[n:].any()  # padding rows are left alone


def _run_kernel(pkg, case, inp, via_module):
    import torch
    qkv = datagen.storage_to_torch(inp["qkv"].copy(), case.dtype).to(DEV)
    q, k, v = qkv.unbind(dim=1)
    out = torch.full((qkv.shape[0], case.num_heads, case.head_size), float("nan"), dtype=qkv.dtype, device=DEV)
    scale = case.head_size ** -0.5
    if via_module:
        pkg.attention_ops.multi_query_kv_attention(out, q, k, v, list(case.seq_lens), scale)
    else:
        pkg._capi.multi_query_kv_attention(out, q, k, v, [int(x) for x in inp["cu_seqlens"]], scale)
    torch.cuda.synchronize()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("via_module", [False, True], ids=["c_abi", "module"])
@pytest.mark.parametrize("name", sorted(META))
def test_kernel_matches_reference_fixture(pkg, name, via_module):
    import torch
    case = _case(name)
    inp = datagen.make_prompt_inputs(case)
    out = _run_kernel(pkg, case, inp, via_module)
    n = int(inp["cu_seqlens"][-1])
    _check_against_fixture(name, case, datagen.torch_to_storage(out, case.dtype), n)
    assert bool(torch.isnan(out[n:].float()).all())  # padding rows are not written


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float16", "bfloat16", "float32"])
@pytest.mark.parametrize("head_size", [64, 80, 96, 128])
def test_kernel_matches_oracle_on_ragged_prompts(pkg, dtype, head_size):
    """Lengths on both sides of the 32 / 64 / 128-row tile edges, several
    workgroups per prompt, three heads, value magnitudes that differ per row."""
    lens = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 257, 5]
    case = datagen.PromptCase(name="ragged", dtype=dtype, seq_lens=lens, num_heads=3, head_size=head_size,
                              dist="normal", pad_tokens=3, seed=900 + head_size)
    inp = datagen.make_prompt_inputs(case)
    out = datagen.torch_to_storage(_run_kernel(pkg, case, inp, False), dtype)
    qkv = inp["qkv"]
    ref = pa_oracle.multi_query_kv_attention(qkv[:, 0], qkv[:, 1], qkv[:, 2], inp["cu_seqlens"],
                                             head_size ** -0.5, dtype)
    n = int(inp["cu_seqlens"][-1])
    got = datagen.storage_to_float32(out, dtype)[:n].astype(np.float64)
    want = datagen.storage_to_float32(ref, dtype)[:n].astype(np.float64)
    bound = EPS[dtype] * (2.0 + 3.0 * np.abs(want)) + 2e-6
    err = np.abs(got - want)
    assert np.all(err <= bound), f"max err {err.max():.3e}"
    # the first token of every prompt attends to itself only: out = v (exactly in 16-bit
    # storage; in fp32 the weight is exp2 of a rounding residual, 1 within a few ulp)
    v = qkv[:, 2]
    for s in inp["cu_seqlens"][:-1]:
        if dtype == "float32":
            assert np.allclose(out[s], v[s], rtol=1e-6, atol=0)
        else:
            assert np.array_equal(out[s], v[s])


@pytest.mark.gpu
def test_kernel_scale_sign_and_zero(pkg):
    """Negative and zero scales take the paths that move the sign into q."""
    import torch
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    T, H, D = 150, 2, 64
    qkv = torch.randn(T, 3, H, D, dtype=torch.float16, device=DEV, generator=g)
    q, k, v = qkv.unbind(dim=1)
    for scale in (-0.125, 0.0, 0.3):
        out = torch.empty(T, H, D, dtype=torch.float16, device=DEV)
        pkg._capi.multi_query_kv_attention(out, q, k, v, [0, 100, 150], scale)
        for lo, hi in ((0, 100), (100, 150)):
            s = torch.einsum("qhd,khd->hqk", q[lo:hi].float(), k[lo:hi].float()) * scale
            s = s + torch.full((hi - lo, hi - lo), float("-inf"), device=DEV).triu(1)
            want = torch.einsum("hqk,khd->qhd", torch.softmax(s, dim=-1), v[lo:hi].float())
            assert torch.allclose(out[lo:hi].float(), want, atol=4e-3, rtol=4e-3), scale


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
def test_kernel_at_full_size_against_torch(pkg, dtype_name):
    """LLaMA-7B shaped prompt batch (32 heads x 128, 2560 tokens, the scheduler's
    max_num_batched_tokens) against torch's attention on the GPU, plus two
    size-independent properties: with V = 1 every output is 1 (the weights sum
    to one), and the result does not depend on how prompts are packed."""
    import torch
    import torch.nn.functional as F
    dtype = getattr(torch, dtype_name)
    g = torch.Generator(device=DEV)
    g.manual_seed(11)
    lens = [1024, 700, 512, 300, 24]
    T, H, D = sum(lens), 32, 128
    qkv = torch.randn(T + 7, 3, H, D, dtype=dtype, device=DEV, generator=g)
    q, k, v = qkv.unbind(dim=1)
    scale = D ** -0.5
    out = torch.zeros(T + 7, H, D, dtype=dtype, device=DEV)
    pkg.attention_ops.multi_query_kv_attention(out, q, k, v, lens, scale)
    lo = 0
    tol = 3e-3 if dtype_name == "float16" else 2e-2
    for n in lens:
        o = F.scaled_dot_product_attention(q[lo:lo + n].transpose(0, 1).float(), k[lo:lo + n].transpose(0, 1).float(),
                                           v[lo:lo + n].transpose(0, 1).float(), is_causal=True, scale=scale)
        assert torch.allclose(out[lo:lo + n].float(), o.transpose(0, 1), atol=tol, rtol=tol)
        lo += n
    assert not out[T:].any()
    # weights sum to one
    ones = torch.ones_like(v)
    out1 = torch.empty(T + 7, H, D, dtype=dtype, device=DEV)
    pkg.attention_ops.multi_query_kv_attention(out1, q, k, ones, lens, scale)
    assert torch.allclose(out1[:T].float(), torch.ones(T, H, D, device=DEV), atol=4e-3 if dtype_name == "float16" else 2e-2)
    # the third prompt alone gives bit-identical rows
    lo = lens[0] + lens[1]
    alone = torch.empty(lens[2], H, D, dtype=dtype, device=DEV)
    pkg.attention_ops.multi_query_kv_attention(alone, q[lo:lo + lens[2]], k[lo:lo + lens[2]], v[lo:lo + lens[2]],
                                               [lens[2]], scale)
    assert torch.equal(alone, out[lo:lo + lens[2]])


def test_prompt_attention_argument_validation(pkg):
    import ctypes
    c = pkg._capi
    one = ctypes.c_void_p(16)
    cu = (ctypes.c_int32 * 3)(0, 4, 9)
    bad = (ctypes.c_int32 * 3)(0, 5, 3)
    f = c.lib.pa_multi_query_kv_attention
    assert f(one, one, one, one, cu, 0, 2, 64, 0.1, 128, 128, 128, 1, None) == c.PA_OK
    assert f(one, one, one, one, bad, 2, 2, 64, 0.1, 128, 128, 128, 1, None) == c.PA_ERR_INVALID_ARGUMENT
    assert f(one, one, one, one, cu, 2, 2, 64, 0.1, 64, 128, 128, 1, None) == c.PA_ERR_INVALID_ARGUMENT
    assert f(one, one, one, one, cu, 2, 2, 64, 0.1, 128, 128, 128, 7, None) == c.PA_ERR_UNSUPPORTED
    assert f(one, one, one, one, cu, 2, 2, 72, 0.1, 144, 144, 144, 1, None) == c.PA_ERR_UNSUPPORTED
    assert "head size" in c.last_error()
    assert f(ctypes.c_void_p(8), one, one, one, cu, 2, 2, 64, 0.1, 128, 128, 128, 1, None) == c.PA_ERR_UNSUPPORTED
    assert f(one, None, one, one, cu, 2, 2, 64, 0.1, 128, 128, 128, 1, None) == c.PA_ERR_INVALID_ARGUMENT
    assert f(one, one, one, one, cu, 2, 2, 64, float("nan"), 128, 128, 128, 1, None) == c.PA_ERR_INVALID_ARGUMENT
    import torch
    x = torch.zeros(4, 2, 64)
    with pytest.raises(RuntimeError, match="must be on the GPU"):
        pkg.attention_ops.multi_query_kv_attention(x.clone(), x, x, x, [4], 0.1)


COMPAT = os.path.join(os.path.dirname(helpers.GOLDEN_DIR), os.pardir, "vllm-v0.1.0_amd", "compat")


def _import_shim(pkg):
    import importlib
    import sys
    path = os.path.abspath(COMPAT)
    if path not in sys.path:
        sys.path.insert(0, path)
    return importlib.import_module("xformers.ops"), importlib.import_module("xformers.ops.fmha.attn_bias")


def test_xformers_import_shim_surface(pkg):
    xops, attn_bias = _import_shim(pkg)
    mask = attn_bias.BlockDiagonalCausalMask.from_seqlens([3, 5])
    assert isinstance(mask, xops.AttentionBias) and mask.seqlens == (3, 5)
    assert xops.fmha.cutlass.FwOp() is not None
    import torch
    x = torch.zeros(1, 8, 2, 64)
    with pytest.raises(NotImplementedError):
        xops.memory_efficient_attention_forward(x, x, x, attn_bias=None)
    with pytest.raises(NotImplementedError):
        xops.memory_efficient_attention_forward(x, x, x, attn_bias=mask, p=0.1)


@pytest.mark.gpu
def test_reference_call_sequence_through_the_import_shim(pkg):
    """The statements of PagedAttention.multi_query_kv_attention
    (vllm/model_executor/layers/attention.py:44, 59-69) and of the reference
    test (tests/kernels/test_attention.py:244-256), run against the shim."""
    import torch
    xops, attn_bias_mod = _import_shim(pkg)
    seq_lens = [9, 130, 77]
    T, H, D = sum(seq_lens), 3, 96
    qkv = torch.empty(T, 3, H, D, dtype=torch.half, device=DEV).uniform_(-1e-3, 1e-3)
    query, key, value = qkv.unbind(dim=1)
    scale = float(1.0 / (D ** 0.5))
    attn_op = xops.fmha.cutlass.FwOp()
    attn_bias = attn_bias_mod.BlockDiagonalCausalMask.from_seqlens(seq_lens)
    out = xops.memory_efficient_attention_forward(query.unsqueeze(0), key.unsqueeze(0), value.unsqueeze(0),
                                                  attn_bias=attn_bias, p=0.0, scale=scale, op=attn_op)
    output = torch.empty(T, H, D, dtype=torch.half, device=DEV)
    output.copy_(out.squeeze(0))
    ref = pa_oracle.multi_query_kv_attention(*(datagen.torch_to_storage(t, "float16") for t in (query, key, value)),
                                             np.concatenate([[0], np.cumsum(seq_lens)]), scale, "float16")
    got = datagen.torch_to_storage(output, "float16").astype(np.float32)
    assert np.allclose(got, ref.astype(np.float32), atol=1e-3, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float16", "bfloat16", "float32"])
def test_kernel_on_scores_that_drive_the_running_maximum(pkg, dtype):
    """Score profiles chosen to exercise the lazily updated running maximum of
    the online softmax: slow ramps (the maximum creeps up by less than the
    rescale threshold per tile), steep ramps (every tile rescales), isolated
    jumps at scattered keys, a maximum at the very first key (never rescales
    again), and large negative scores whose weights underflow.  One head per
    profile; q = a fixed direction, k_j = that direction times the profile."""
    import torch
    D, n = 128, 900
    rng = np.random.default_rng(3)
    j = np.arange(n, dtype=np.float64)
    profiles = [
        0.02 * j,                                   # +1.3 per 64 keys: long deferral
        0.2 * j,                                    # +13 per tile: rescale every tile
        np.where(rng.random(n) < 0.02, 25.0, 0.0) + rng.normal(0, 0.3, n),   # isolated jumps
        -0.05 * j,                                  # maximum at key 0
        np.where(j % 97 == 0, 0.0, -80.0),          # everything else underflows
        rng.normal(0, 4.0, n),                      # plain noise, wide
    ]
    H = len(profiles)
    scale = D ** -0.5
    u = rng.normal(0, 1, D)
    u /= np.linalg.norm(u)
    q = np.zeros((n, H, D))
    k = np.zeros((n, H, D))
    for h, prof in enumerate(profiles):
        q[:, h] = u * 4.0                           # q.k = 4 * amp_j ; times scale -> profile
        k[:, h] = u[None, :] * (prof[:, None] / (4.0 * scale))
    v = rng.normal(0, 1, (n, H, D))
    qkv = datagen.to_storage(np.stack([q, k, v], axis=1).reshape(-1), dtype).reshape(n, 3, H, D)
    lens = [n - 300, 300]
    cu = np.array([0, lens[0], n], dtype=np.int32)
    ref = pa_oracle.multi_query_kv_attention(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu, scale, dtype)
    t = datagen.storage_to_torch(qkv.copy(), dtype).to(DEV)
    out = torch.empty(n, H, D, dtype=t.dtype, device=DEV)
    pkg._capi.multi_query_kv_attention(out, t[:, 0], t[:, 1], t[:, 2], [int(x) for x in cu], scale)
    torch.cuda.synchronize()
    got = datagen.storage_to_float32(datagen.torch_to_storage(out, dtype), dtype).astype(np.float64)
    want = datagen.storage_to_float32(ref, dtype).astype(np.float64)
    assert np.all(np.isfinite(got))
    bound = EPS[dtype] * (2.0 + 3.0 * np.abs(want)) + 2e-6
    err = np.abs(got - want)
    assert np.all(err <= bound), f"max err {err.max():.3e} at head {np.unravel_index(err.argmax(), err.shape)[1]}"


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name,H,D,lens", [
    ("float16", 32, 128, [1024] * 8),
    ("bfloat16", 32, 80, [700] * 11 + [1]),
    ("float16", 40, 64, [300, 513, 256, 1000, 255, 257, 2048, 31, 777]),
    ("float32", 32, 96, [520] * 11),
    ("bfloat16", 32, 64, [2048] * 4 + [100]),   # head size 64: two 8-wave workgroups share a CU
])
def test_kernel_with_256_row_workgroups_against_torch(pkg, dtype_name, H, D, lens):
    """Batches big enough for the 8-wave (256 query rows) workgroups, with
    lengths on both sides of their edges, against torch's attention."""
    import torch
    import torch.nn.functional as F
    assert sum((n + 255) // 256 for n in lens) * H >= 1024  # the library's threshold for 8 waves
    dtype = getattr(torch, dtype_name)
    g = torch.Generator(device=DEV)
    g.manual_seed(21)
    T = sum(lens)
    qkv = torch.randn(T, 3, H, D, dtype=dtype, device=DEV, generator=g)
    q, k, v = qkv.unbind(dim=1)
    scale = D ** -0.5
    out = torch.full((T, H, D), float("nan"), dtype=dtype, device=DEV)
    pkg.attention_ops.multi_query_kv_attention(out, q, k, v, lens, scale)
    tol = {"float16": 3e-3, "bfloat16": 2e-2, "float32": 1e-4}[dtype_name]
    lo = 0
    for n in lens:
        o = F.scaled_dot_product_attention(q[lo:lo + n].transpose(0, 1).float(), k[lo:lo + n].transpose(0, 1).float(),
                                           v[lo:lo + n].transpose(0, 1).float(), is_causal=True, scale=scale)
        assert torch.allclose(out[lo:lo + n].float(), o.transpose(0, 1), atol=tol, rtol=tol), n
        lo += n


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_kernel_randomized_shapes_against_oracle(pkg, seed):
    """Differential test over random prompt batches: number and lengths of
    prompts, head count, head size, dtype, scale, leading and trailing rows
    that belong to no prompt (cu_seqlens[0] > 0)."""
    import torch
    rng = np.random.default_rng(3000 + seed)
    dtype = ["float16", "bfloat16", "float32"][seed % 3]
    D = int(rng.choice([64, 80, 96, 128]))
    H = int(rng.choice([1, 2, 3, 4]))
    lens = [int(x) for x in rng.integers(1, 400, size=int(rng.integers(1, 7)))]
    lead, trail = int(rng.integers(0, 9)), int(rng.integers(0, 9))
    case = datagen.PromptCase(name=f"fuzz{seed}", dtype=dtype, seq_lens=[lead] + lens, num_heads=H, head_size=D,
                              dist="normal", pad_tokens=trail, seed=3100 + seed)
    inp = datagen.make_prompt_inputs(case)
    cu = inp["cu_seqlens"][1:]  # the first `lead` rows are outside every prompt
    scale = float(rng.uniform(0.03, 0.3))
    qkv = inp["qkv"]
    ref = pa_oracle.multi_query_kv_attention(qkv[lead:, 0], qkv[lead:, 1], qkv[lead:, 2], cu - lead, scale, dtype)
    t = datagen.storage_to_torch(qkv.copy(), dtype).to(DEV)
    out = torch.full((qkv.shape[0], H, D), float("nan"), dtype=t.dtype, device=DEV)
    pkg._capi.multi_query_kv_attention(out, t[:, 0], t[:, 1], t[:, 2], [int(x) for x in cu], scale)
    torch.cuda.synchronize()
    n = int(cu[-1])
    assert bool(torch.isnan(out[:lead].float()).all()) and bool(torch.isnan(out[n:].float()).all())
    got = datagen.storage_to_float32(datagen.torch_to_storage(out[lead:n], dtype), dtype).astype(np.float64)
    want = datagen.storage_to_float32(ref, dtype)[: n - lead].astype(np.float64)
    bound = EPS[dtype] * (2.0 + 3.0 * np.abs(want)) + 2e-6
    err = np.abs(got - want)
    assert np.all(err <= bound), f"seed {seed} lens={lens} H={H} D={D} {dtype}: max err {err.max():.3e}"


@pytest.mark.gpu
def test_kernel_with_more_prompts_than_one_launch_carries(pkg):
    """The prompt table rides in the kernel arguments, 256 prompts per launch: a batch of 300 short prompts
    is served by two launches of one call; every prompt against torch's attention, rows past the batch untouched."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    lens = [int(x) for x in rng.integers(1, 48, size=300)]
    T, H, D = sum(lens), 2, 64
    g = torch.Generator(device=DEV)
    g.manual_seed(31)
    qkv = torch.randn(T + 3, 3, H, D, dtype=torch.float16, device=DEV, generator=g)
    q, k, v = qkv.unbind(dim=1)
    out = torch.full((T + 3, H, D), float("nan"), dtype=torch.float16, device=DEV)
    pkg.attention_ops.multi_query_kv_attention(out, q, k, v, lens, D ** -0.5)
    lo = 0
    for n in lens:
        o = F.scaled_dot_product_attention(q[lo:lo + n].transpose(0, 1).float(), k[lo:lo + n].transpose(0, 1).float(),
                                           v[lo:lo + n].transpose(0, 1).float(), is_causal=True, scale=D ** -0.5)
        assert torch.allclose(out[lo:lo + n].float(), o.transpose(0, 1), atol=3e-3, rtol=3e-3), (lo, n)
        lo += n
    assert torch.isnan(out[T:]).all()


@pytest.mark.gpu
def test_kernel_with_more_query_blocks_than_one_launch_carries(pkg):
    """One prompt of 66 000 tokens is 516 query blocks of 128 rows, more than the 512 work items a launch
    carries: the call splits into two launches.  Too long for a dense reference: with V = 1 every output is 1,
    and sampled rows (first, last, both sides of the launches' seam) equal the directly computed attention."""
    import torch
    n, H, D = 66000, 1, 64
    assert (n + 127) // 128 > 512 and (n + 255) // 256 * H < 1024  # 128-row workgroups, two launches
    g = torch.Generator(device=DEV)
    g.manual_seed(32)
    qkv = torch.randn(n, 3, H, D, dtype=torch.bfloat16, device=DEV, generator=g)
    q, k, v = qkv.unbind(dim=1)
    scale = D ** -0.5
    out = torch.full((n, H, D), float("nan"), dtype=torch.bfloat16, device=DEV)
    pkg.attention_ops.multi_query_kv_attention(out, q, k, v, [n], scale)
    assert torch.isfinite(out.float()).all()
    rows = [0, 1, 127, 128, 4 * 128 - 1, 4 * 128, 4 * 128 + 1, 33000, n - 129, n - 128, n - 1]
    kf, vf = k[:, 0].float(), v[:, 0].float()
    for i in rows:
        w = torch.softmax((kf[:i + 1] @ q[i, 0].float()) * scale, dim=0)
        want = w @ vf[:i + 1]
        assert torch.allclose(out[i, 0].float(), want, atol=2e-2, rtol=2e-2), i
    ones = torch.ones_like(v)
    pkg.attention_ops.multi_query_kv_attention(out, q, k, ones, [n], scale)
    assert torch.allclose(out.float(), torch.ones(n, H, D, device=DEV), atol=2e-2)
