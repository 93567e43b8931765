"""GPU parity tests of the one-launch form of a decode step's cache write + attention
(pa_append_single_query_cached_kv_attention), through the C ABI and the `attention_ops` module.

Bar: `out` and both caches bit-identical to reshape_and_cache followed by the decode entry point on
the same arguments (the generation branch of vllm/model_executor/layers/attention.py:134-161), and
both within the decode tolerance of the C oracle's reshape_and_cache + attention."""
import numpy as np
import pytest

from oracle import datagen, pa_oracle
from tests import helpers

pytestmark = pytest.mark.gpu


def _slots(inp, B):
    cl, bt = inp["context_lens"].astype(np.int64), inp["block_tables"].astype(np.int64)
    pos = cl - 1
    return (bt[np.arange(len(cl)), pos // B] * B + pos % B).astype(np.int32)


def _bits(t):
    import torch
    return t.contiguous().view(torch.int16 if t.element_size() == 2 else torch.int32)


def _both_ways(pkg, case, form, total_hint=False, device="cuda:0"):
    """Returns (inp, two-call results, one-launch results), each (out, key_cache, value_cache)."""
    import torch
    c = pkg._capi
    inp = datagen.make_decode_inputs(case)
    B = case.block_size
    qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, device)
    key, value = qkv[:, 1], qkv[:, 2]
    slots = torch.from_numpy(_slots(inp, B)).to(device)
    mc = inp["max_context_len"]
    total = int(inp["context_lens"].sum()) if total_hint else -1
    nbytes = c.decode_workspace_bytes(case.num_seqs, case.num_heads, case.head_size, B, mc) if form != "rows" else 0
    results = []
    for fused in (False, True):
        k2, v2 = kc.clone(), vc.clone()
        out = torch.full((case.num_seqs, case.num_heads, case.head_size), float("nan"), dtype=qkv.dtype, device=device)
        ws = torch.full((max(nbytes, 4) // 4,), float("nan"), dtype=torch.float32, device=device) if nbytes else None
        if fused:
            c.append_single_query_cached_kv_attention(out, query, key, value, k2, v2, inp["scale"], bt, cl, B, mc, ws,
                                                      -1 if form == "per_call" else 0, total)
        else:
            c.reshape_and_cache(key, value, k2, v2, slots)
            if form == "rows":
                c.single_query_cached_kv_attention(out, query, k2, v2, inp["scale"], bt, cl, B, mc)
            elif form == "per_call":
                c.single_query_cached_kv_attention_ws(out, query, k2, v2, inp["scale"], bt, cl, B, mc, ws)
            else:
                c.single_query_cached_kv_attention_planned(out, query, k2, v2, inp["scale"], bt, cl, B, mc, ws, False, total)
        torch.cuda.synchronize()
        results.append((out, k2, v2))
    return inp, results[0], results[1]


def _assert_identical(two, one, what):
    import torch
    for name, a, b in zip(("out", "key_cache", "value_cache"), two, one):
        assert torch.equal(_bits(a), _bits(b)), f"{what}: {name} differs from the two-call sequence"


def _assert_oracle(inp, case, one, what):
    B = case.block_size
    kc, vc = inp["key_cache"].copy(), inp["value_cache"].copy()
    pa_oracle.reshape_and_cache(inp["qkv"][:, 1], inp["qkv"][:, 2], kc, vc, _slots(inp, B), case.dtype)
    ref = pa_oracle.single_query_cached_kv_attention(inp["qkv"][:, 0], kc, vc, inp["scale"], inp["block_tables"],
                                                     inp["context_lens"], case.dtype)
    out, k2, v2 = one
    assert np.array_equal(datagen.torch_to_storage(k2.cpu(), case.dtype), kc), f"{what}: key cache vs oracle"
    assert np.array_equal(datagen.torch_to_storage(v2.cpu(), case.dtype), vc), f"{what}: value cache vs oracle"
    inp2 = dict(inp, key_cache=kc, value_cache=vc)
    atol, rtol = helpers.decode_atol(inp2, case.dtype), helpers.TIGHT[case.dtype][1]
    helpers.assert_close(helpers.out_to_f32(out, case.dtype), datagen.storage_to_float32(ref, case.dtype), atol, 2 * rtol, what)


@pytest.mark.parametrize("form", ["rows", "planned", "per_call"])
@pytest.mark.parametrize("dtype", datagen.DTYPES)
@pytest.mark.parametrize("D,B", [(64, 8), (64, 16), (80, 16), (96, 32), (128, 8), (128, 16), (128, 32)])
def test_append_decode_ragged_batch(pkg, dtype, D, B, form):
    """Ragged batch, every tile shape and data type, all three ways of planning; small batches take
    uniform partitions in the planned forms."""
    case = datagen.DecodeCase(name="app", dtype=dtype, num_seqs=9, num_heads=3, head_size=D, block_size=B,
                              num_blocks=9 * (700 // B + 1), max_context=700, dist="normal", tables="permutation",
                              seed=100 + D + B)
    inp, two, one = _both_ways(pkg, case, form)
    _assert_identical(two, one, f"{dtype} {D}x{B} {form}")
    _assert_oracle(inp, case, one, f"{dtype} {D}x{B} {form} vs oracle")


@pytest.mark.parametrize("ctxs", [[1], [1, 2, 15, 16, 17, 31, 32, 33], [16, 32, 48, 64], [255, 256, 257, 1, 129]])
def test_append_decode_block_boundaries(pkg, ctxs):
    """First token of a sequence, first and last slot of a block, exact multiples of the block size."""
    case = datagen.DecodeCase(name="edge", dtype="float16", num_seqs=len(ctxs), num_heads=4, head_size=128, block_size=16,
                              num_blocks=128, context_lens=ctxs, dist="normal", tables="permutation", seed=7)
    for form in ("rows", "planned"):
        inp, two, one = _both_ways(pkg, case, form)
        _assert_identical(two, one, f"ctxs {ctxs} {form}")
        _assert_oracle(inp, case, one, f"ctxs {ctxs} {form} vs oracle")


def test_append_decode_item_list_and_balanced_order(pkg):
    """Big batch: balanced row order; long contexts among short ones: item list with and without the
    total-length hint; bfloat16 and float16."""
    for dtype, S, H, maxc, hint in (("float16", 160, 8, 900, False), ("bfloat16", 72, 8, 3000, True),
                                     ("float16", 72, 8, 3000, False)):
        case = datagen.DecodeCase(name="big", dtype=dtype, num_seqs=S, num_heads=H, head_size=128, block_size=16,
                                  num_blocks=S * ((maxc + 15) // 16) + 8, max_context=maxc, dist="normal",
                                  tables="permutation", seed=55 + S)
        inp, two, one = _both_ways(pkg, case, "planned", total_hint=hint)
        _assert_identical(two, one, f"{dtype} S{S} ctx<={maxc} hint={hint}")
        _assert_oracle(inp, case, one, f"{dtype} S{S} ctx<={maxc} hint={hint} vs oracle")


def test_append_decode_forked_tables(pkg):
    """Sequences that share their prompt's full blocks: the last block of each is its own."""
    case = datagen.DecodeCase(name="fork", dtype="float16", num_seqs=12, num_heads=16, head_size=96, block_size=16,
                              num_blocks=600, max_context=500, dist="normal", tables="forked", fork=3, seed=77)
    for form in ("rows", "planned"):
        inp, two, one = _both_ways(pkg, case, form)
        _assert_identical(two, one, f"forked {form}")
        _assert_oracle(inp, case, one, f"forked {form} vs oracle")


def test_append_decode_shared_plan_over_layers(pkg):
    """plan_ready = 0 on the first layer, 1 on the others, as a decode step uses it; then a second step
    whose contexts are one token longer."""
    import torch
    c = pkg._capi
    S, H, D, B, L = 96, 8, 128, 16, 3
    # no context ends a block: the second step's token then still fits the blocks the tables hold
    ctxs = [int(v) - (int(v) % B == 0) for v in datagen.randint(S, 910, 2, 600)]
    case = datagen.DecodeCase(name="step", dtype="float16", num_seqs=S, num_heads=H, head_size=D, block_size=B,
                              num_blocks=S * 40, context_lens=ctxs, dist="normal", tables="permutation", seed=91)
    inp = datagen.make_decode_inputs(case)
    qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, "cuda:0")
    key, value = qkv[:, 1], qkv[:, 2]
    mc = inp["max_context_len"] + 1
    nbytes = c.decode_workspace_bytes(S, H, D, B, mc)
    ws = torch.zeros(max(nbytes, 4), dtype=torch.uint8, device="cuda:0")
    caches = [[(kc.clone(), vc.clone()) for _ in range(L)] for _ in range(2)]
    ctx = inp["context_lens"].copy()
    for step in range(2):
        slots = torch.from_numpy(_slots(dict(inp, context_lens=ctx), B)).cuda()
        cl_t = torch.from_numpy(ctx).cuda()
        total = int(ctx.sum())
        for layer in range(L):
            outs = []
            for fused in (False, True):
                k2, v2 = caches[fused][layer]
                out = torch.empty(S, H, D, dtype=qkv.dtype, device="cuda:0")
                if fused:
                    c.append_single_query_cached_kv_attention(out, query, key, value, k2, v2, inp["scale"], bt, cl_t, B,
                                                              mc, ws, 1 if layer else 0, total)
                else:
                    c.reshape_and_cache(key, value, k2, v2, slots)
                    c.single_query_cached_kv_attention_planned(out, query, k2, v2, inp["scale"], bt, cl_t, B, mc, ws,
                                                               False, total)
                torch.cuda.synchronize()
                outs.append(out)
            _assert_identical((outs[0],) + caches[0][layer], (outs[1],) + caches[1][layer], f"step {step} layer {layer}")
        ctx = ctx + 1


def test_append_decode_via_attention_ops_module(pkg):
    import torch
    case = datagen.DecodeCase(name="mod", dtype="bfloat16", num_seqs=37, num_heads=3, head_size=80, block_size=16,
                              num_blocks=37 * 19, max_context=300, dist="normal", tables="permutation", seed=13)
    inp = datagen.make_decode_inputs(case)
    qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, "cuda:0")
    out = torch.empty(37, 3, 80, dtype=qkv.dtype, device="cuda:0")
    ws = torch.empty(pkg.attention_ops.decode_workspace_bytes(37, 3, 80, 16, inp["max_context_len"]), dtype=torch.uint8,
                     device="cuda:0")
    pkg.attention_ops.reshape_and_cache_single_query_cached_kv_attention(
        out, query, qkv[:, 1], qkv[:, 2], kc, vc, inp["scale"], bt, cl, 16, inp["max_context_len"], ws, False, -1)
    torch.cuda.synchronize()
    _assert_oracle(inp, case, (out, kc, vc), "attention_ops module")
    with pytest.raises(RuntimeError, match="shapes differ"):
        pkg.attention_ops.reshape_and_cache_single_query_cached_kv_attention(
            out, query, qkv[:5, 1], qkv[:, 2], kc, vc, inp["scale"], bt, cl, 16, inp["max_context_len"], ws, False, -1)


def test_append_decode_rejects_short_token_rows(pkg):
    import torch
    c = pkg._capi
    S, H, D, B = 4, 2, 64, 16
    q = torch.zeros(S, H, D, dtype=torch.float16, device="cuda:0")
    kc = torch.zeros(8, H, D // 8, B, 8, dtype=torch.float16, device="cuda:0")
    vc = torch.zeros(8, H, D, B, dtype=torch.float16, device="cuda:0")
    bt = torch.arange(8, dtype=torch.int32, device="cuda:0").reshape(S, 2)
    cl = torch.full((S,), 20, dtype=torch.int32, device="cuda:0")
    short = torch.zeros(S, H * D // 2, dtype=torch.float16, device="cuda:0")

    class Rows:  # a [S, H, D] view whose row stride is too short for H x D elements
        shape = (S, H, D)
        dtype = torch.float16

        def data_ptr(self):
            return short.data_ptr()

        def stride(self, i):
            return H * D // 2

    with pytest.raises(pkg._capi.PagedAttnError, match="shorter than a token row"):
        c.append_single_query_cached_kv_attention(q.clone(), q, Rows(), q, kc, vc, 0.1, bt, cl, B, 20)


def test_layer_takes_the_append_form_for_small_generation_batches_only(pkg):
    """PagedAttention.forward: same output and caches whether the generation tokens' cache write rides in the
    attention launch (metadata whose slots are known to follow the tables, small batch) or is a launch of
    its own; big batches and metadata without that promise keep the two launches."""
    import types
    import torch
    from vllm_amd.input_metadata import InputMetadata
    from vllm_amd.layers import PagedAttention
    calls = []
    ops = types.SimpleNamespace(**{n: getattr(pkg.attention_ops, n) for n in dir(pkg.attention_ops) if not n.startswith("_")})
    real = ops.reshape_and_cache_single_query_cached_kv_attention

    def counted(*args):
        calls.append(args[1].shape[0])
        return real(*args)
    ops.reshape_and_cache_single_query_cached_kv_attention = counted
    for S, H, expect_append in ((6, 4, True), (200, 4, False)):
        case = datagen.DecodeCase(name="layer", dtype="float16", num_seqs=S, num_heads=H, head_size=64, block_size=16,
                                  num_blocks=S * 20, max_context=300, dist="normal", tables="permutation", seed=3 + S)
        inp = datagen.make_decode_inputs(case)
        qkv, _, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, "cuda:0")
        q, k, v = qkv.view(S, 3 * H * 64).chunk(3, dim=-1)
        slots = torch.from_numpy(_slots(inp, 16)).cuda()
        layer = PagedAttention(H, 64, inp["scale"], attention_ops=ops)
        results = []
        for promise in (False, True):
            meta = InputMetadata([], {}, [], slots, cl, inp["max_context_len"], bt)
            if promise:
                meta.generation_slots_follow_tables = True
            k2, v2 = kc.clone(), vc.clone()
            n = len(calls)
            out = layer(q, k, v, k2, v2, meta, None)
            torch.cuda.synchronize()
            assert (len(calls) > n) == (promise and expect_append), (S, promise)
            results.append((out, k2, v2))
        _assert_identical(results[0], results[1], f"layer, {S} sequences")
        _assert_oracle(inp, case, (results[1][0].view(S, H, 64), results[1][1], results[1][2]), f"layer, {S} sequences vs oracle")
