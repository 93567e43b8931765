"""rotary_embedding_neox, rms_norm, silu_and_mul: the C oracle against fixtures
produced by the reference's own Python statements (CPU), and the HIP kernels
against both (GPU, through the C ABI and the operator modules).

Bars: rotary is bit-exact (every operation rounds to the storage type, as in
the reference).  rms_norm and silu_and_mul may differ from another correct
implementation by one unit in the last place of the storage type on isolated
elements (fp32 summation order of the variance; last-bit differences between
exp implementations), never more."""
import json
import os

import numpy as np
import pytest

from oracle import datagen, pa_oracle
from tests import helpers

with open(os.path.join(helpers.GOLDEN_DIR, "elementwise_golden.json")) as _f:
    META = json.load(_f)
DATA = np.load(os.path.join(helpers.GOLDEN_DIR, "elementwise_golden.npz"), allow_pickle=False)
DEV = "cuda:0"
ULP = {"float32": 2.0 ** -23, "float16": 2.0 ** -10, "bfloat16": 2.0 ** -7}


def _case(name):
    return datagen.ElementwiseCase(**META[name]["case"])


def _assert_within_ulps(got, want, dtype, what, ulps=1.01, max_fraction=1e-3):
    g = datagen.storage_to_float32(got, dtype).astype(np.float64)
    w = datagen.storage_to_float32(want, dtype).astype(np.float64)
    assert g.shape == w.shape and np.all(np.isfinite(g)), what
    tol = ulps * ULP[dtype] * np.maximum(np.abs(w), 2.0 ** -14)
    assert np.all(np.abs(g - w) <= tol), f"{what}: max err {np.abs(g - w).max()}"
    frac = float(np.mean(g != w))
    assert frac <= max_fraction, f"{what}: {frac:.2e} of the elements differ in the last place"


def _bar(case):
    """fp32 storage: the summation order of the variance and the last bit of exp
    move most results by an ulp or two, so only the size of the error is bounded.
    16-bit storage: the fp32 differences survive rounding on isolated elements."""
    if case.dtype == "float32":
        return dict(ulps=4.01, max_fraction=1.0)
    return dict(ulps=1.01, max_fraction=2e-3)


def _oracle(case, inp):
    if case.op == "rotary":
        q, k = inp["query"].copy(), inp["key"].copy()
        pa_oracle.rotary_embedding_neox(inp["positions"], q, k, case.head_size, inp["cos_sin_cache"], case.dtype)
        return {"query": q, "key": k}
    if case.op == "rms_norm":
        return {"out": pa_oracle.rms_norm(inp["input"], inp["weight"], 1e-6, case.dtype)}
    return {"out": pa_oracle.silu_and_mul(inp["input"], case.dtype)}


def _check(case, got, name, against_gpu=False):
    for key, arr in got.items():
        want = DATA[f"{name}/{key}"]
        if case.op == "rotary":
            assert np.array_equal(arr, want), f"{name}/{key} differs from the reference fixture"
        else:
            _assert_within_ulps(arr, want, case.dtype, f"{name}/{key}", **_bar(case))


@pytest.mark.parametrize("name", sorted(META))
def test_oracle_matches_reference_fixture(name):
    case = _case(name)
    inp = datagen.make_elementwise_inputs(case)
    assert datagen.checksum_arrays(inp) == META[name]["inputs_sha256"]
    _check(case, _oracle(case, inp), name)


def _to_dev(a, dtype):
    return datagen.storage_to_torch(np.ascontiguousarray(a).copy(), dtype).to(DEV)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(META))
def test_kernels_match_reference_fixture_and_oracle(pkg, name):
    import torch
    case = _case(name)
    inp = datagen.make_elementwise_inputs(case)
    dt = case.dtype
    if case.op == "rotary":
        q, k = _to_dev(inp["query"], dt), _to_dev(inp["key"], dt)
        pkg._capi.rotary_embedding_neox(torch.from_numpy(inp["positions"]).to(DEV), q, k, case.head_size,
                                        _to_dev(inp["cos_sin_cache"], dt))
        got = {"query": datagen.torch_to_storage(q, dt), "key": datagen.torch_to_storage(k, dt)}
    elif case.op == "rms_norm":
        x = _to_dev(inp["input"], dt)
        out = torch.empty_like(x)
        pkg._capi.rms_norm(out, x, _to_dev(inp["weight"], dt), 1e-6)
        got = {"out": datagen.torch_to_storage(out, dt)}
    else:
        x = _to_dev(inp["input"], dt)
        out = torch.empty(x.shape[0], x.shape[1] // 2, dtype=x.dtype, device=DEV)
        pkg._capi.silu_and_mul(out, x)
        got = {"out": datagen.torch_to_storage(out, dt)}
    torch.cuda.synchronize()
    _check(case, got, name)
    ref = _oracle(case, inp)
    for key in got:
        if case.op == "rotary":
            assert np.array_equal(got[key], ref[key]), f"{name}/{key} differs from the oracle"
        else:
            _assert_within_ulps(got[key], ref[key], dt, f"{name}/{key} vs oracle", **_bar(case))


@pytest.mark.gpu
def test_operator_modules_like_the_reference_tests(pkg):
    """tests/kernels/test_pos_encoding.py:80-131, test_layernorm.py:25-43,
    test_activation.py:13-24 at their own sizes, through the pybind11 modules,
    with the reference tests' tolerances, against torch on the GPU."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator(device=DEV)
    g.manual_seed(0)
    for dtype in (torch.half, torch.bfloat16, torch.float):
        # rotary: 2145 tokens, 5 heads, q and k are slices of a fused qkv row (llama.py:119)
        for head_size in (32, 64, 80, 96, 128, 160, 192, 256):  # test_pos_encoding.py:136
            T, H, mp = 2145, 5, 8192
            pos = torch.randint(0, mp, (T,), device=DEV, generator=g)
            qkv = torch.randn(T, 3 * H * head_size, dtype=dtype, device=DEV, generator=g)
            q, k, _ = qkv.chunk(3, dim=-1)
            inv = 1.0 / (10000 ** (torch.arange(0, head_size, 2, device=DEV) / head_size))
            fr = torch.einsum("i,j->ij", torch.arange(mp, device=DEV).float(), inv.float())
            cache = torch.cat((fr.cos(), fr.sin()), dim=-1).to(dtype)
            q0, k0 = q.clone(), k.clone()
            pkg.pos_encoding_ops.rotary_embedding_neox(pos, q, k, head_size, cache)
            c, s = cache[pos][:, None, : head_size // 2], cache[pos][:, None, head_size // 2:]
            for got, src in ((q, q0), (k, k0)):
                x = src.view(T, H, head_size)
                x1, x2 = x[..., : head_size // 2], x[..., head_size // 2:]
                want = torch.cat((x1 * c - x2 * s, x2 * c + x1 * s), dim=-1).view(T, H * head_size)
                assert torch.allclose(got, want, atol=1e-3, rtol=1e-5), (dtype, head_size)
        # rms_norm
        # the reference's grid (test_layernorm.py:47-49) and one row longer than the register-resident limit
        for T, hidden in [(T, h) for T in (7, 128, 2048) for h in (13, 64, 1024, 5120)] + [(33, 8200)]:
            x = torch.randn(T, hidden, dtype=dtype, device=DEV, generator=g)
            w = (torch.rand(hidden, device=DEV, generator=g) * 2e-3 - 1e-3).to(dtype)
            out = torch.empty_like(x)
            pkg.layernorm_ops.rms_norm(out, x, w, 1e-6)
            var = x.float().pow(2).mean(-1, keepdim=True)
            want = w * (x * torch.rsqrt(var + 1e-6)).to(dtype)
            assert torch.allclose(out, want, atol=1e-3, rtol=1e-5), (dtype, T, hidden)
        # silu_and_mul
        for T, d in [(T, d) for T in (7, 83, 2048) for d in (512, 4096, 5120, 13824)]:  # test_activation.py:30-31
            x = torch.randn(T, 2 * d, dtype=dtype, device=DEV, generator=g)
            out = torch.empty(T, d, dtype=dtype, device=DEV)
            pkg.activation_ops.silu_and_mul(out, x)
            x1, x2 = x.chunk(2, dim=1)
            want = F.silu(x1) * x2
            tol = {torch.half: 2e-3, torch.bfloat16: 2e-2, torch.float: 1e-5}[dtype]
            assert torch.allclose(out, want, atol=1e-5, rtol=tol), (dtype, T, d)
            if dtype != torch.float:  # isolated last-place differences only
                assert float((out != want).float().mean()) < 2e-3
    torch.cuda.synchronize()


def test_elementwise_argument_validation(pkg):
    import ctypes
    c = pkg._capi
    one = ctypes.c_void_p(16)
    assert c.lib.pa_rotary_embedding_neox(one, one, one, one, 4, 2, 64, 66, 128, 128, 1, None) == c.PA_ERR_INVALID_ARGUMENT
    assert c.lib.pa_rotary_embedding_neox(one, one, one, one, 4, 2, 64, 31, 128, 128, 1, None) == c.PA_ERR_INVALID_ARGUMENT
    assert c.lib.pa_rotary_embedding_neox(one, one, one, one, 0, 2, 64, 64, 128, 128, 1, None) == c.PA_OK
    assert c.lib.pa_rotary_embedding_neox(one, one, one, one, 4, 2, 64, 64, 128, 128, 5, None) == c.PA_ERR_UNSUPPORTED
    assert c.lib.pa_rms_norm(one, None, one, 1e-6, 3, 8, 1, None) == c.PA_ERR_INVALID_ARGUMENT
    assert c.lib.pa_rms_norm(one, one, one, 1e-6, 0, 8, 1, None) == c.PA_OK
    assert c.lib.pa_silu_and_mul(None, one, 3, 8, 1, None) == c.PA_ERR_INVALID_ARGUMENT
    assert c.lib.pa_silu_and_mul(one, one, 3, 8, 3, None) == c.PA_ERR_UNSUPPORTED
    import torch
    x = torch.zeros(2, 8)
    with pytest.raises(RuntimeError, match="must be on the GPU"):
        pkg.activation_ops.silu_and_mul(torch.zeros(2, 4), x)
    with pytest.raises(RuntimeError, match="must be on the GPU"):
        pkg.layernorm_ops.rms_norm(torch.zeros(2, 8), x, torch.zeros(8), 1e-6)
    with pytest.raises(RuntimeError, match="must be on the GPU"):
        pkg.pos_encoding_ops.rotary_embedding_neox(torch.zeros(2, dtype=torch.long), x, x.clone(), 8, torch.zeros(4, 8))
