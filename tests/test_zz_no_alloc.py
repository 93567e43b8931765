"""pa_single_query_cached_kv_attention_static: decode attention on the
library's own per-GPU scratch array, for callers that cannot allocate.  Checked
through the C ABI against pa_single_query_cached_kv_attention_ws (same kernels,
same launch plan: bit-identical results) and against the oracle, on calls in a
row that share the array — each launch form, then a smaller input."""
import numpy as np
import pytest

from oracle import datagen, pa_oracle
from tests import helpers

pytestmark = pytest.mark.gpu


def _case(name, S, H, D, B, ctx, seed, tables="permutation", dtype="float16"):
    return datagen.DecodeCase(name=name, dtype=dtype, num_seqs=S, num_heads=H, head_size=D, block_size=B,
                              num_blocks=sum((c + B - 1) // B for c in ctx) + 4, context_lens=list(ctx), dist="normal",
                              tables=tables, seed=seed)


def _both(pkg, case):
    """(static-scratch result, _ws result with a caller workspace, oracle) of one case."""
    import torch
    dev = "cuda:0"
    inp = datagen.make_decode_inputs(case)
    qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, dev)
    B, mc = case.block_size, inp["max_context_len"]
    shape = (case.num_seqs, case.num_heads, case.head_size)
    out_static = torch.full(shape, float("nan"), dtype=qkv.dtype, device=dev)
    out_ws = torch.full(shape, float("nan"), dtype=qkv.dtype, device=dev)
    nbytes = pkg._capi.decode_workspace_bytes(case.num_seqs, case.num_heads, case.head_size, B, mc)
    ws = torch.full((max(nbytes, 4) // 4,), float("nan"), dtype=torch.float32, device=dev) if nbytes else None
    torch.cuda.synchronize()
    # The entry point is reached through ctypes, so torch's allocator statistics would not see a
    # hipMalloc inside the library: the check is at the driver.  (1) hipMemGetInfo's free bytes do
    # not move across the call — after one warm-up call on this device, which loads the code object
    # and resolves the scratch symbol's address; torch's counters are watched too, so that a change
    # of the free bytes cannot be an allocation of torch's own.  (2) The call is captured into a HIP
    # graph in the capture mode in which hipMalloc, hipFree and every synchronising call are errors,
    # and the replay gives the same bits.
    def handed_out():
        st = torch.cuda.memory_stats()
        return st["allocation.all.allocated"], st["segment.all.allocated"]

    def call(out):
        pkg._capi.single_query_cached_kv_attention_static(out, query, kc, vc, inp["scale"], bt, cl, B, mc)
    call(out_static)  # warm-up
    torch.cuda.synchronize()
    out_static.fill_(float("nan"))
    torch.cuda.synchronize()
    before, free_before = handed_out(), torch.cuda.mem_get_info()[0]
    call(out_static)
    torch.cuda.synchronize()
    assert handed_out() == before
    assert torch.cuda.mem_get_info()[0] == free_before, "device memory was allocated or freed inside the call"
    out_graph = torch.full(shape, float("nan"), dtype=qkv.dtype, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side, capture_error_mode="global"):
        pkg._capi.single_query_cached_kv_attention_static(out_graph, query, kc, vc, inp["scale"], bt, cl, B, mc,
                                                          stream=side)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_graph, out_static), "the captured call replays to other bits than the eager call"
    pkg._capi.single_query_cached_kv_attention_ws(out_ws, query, kc, vc, inp["scale"], bt, cl, B, mc, ws)
    torch.cuda.synchronize()
    ref = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], case.dtype), case.dtype)
    return out_static, out_ws, ref, nbytes


def test_static_scratch_decode_matches_the_workspace_form_on_consecutive_calls(pkg):
    import torch
    rng = np.random.default_rng(3)
    big_ragged = [int(c) for c in rng.integers(1, 700, size=80)]
    long_mixed = [2600 if i % 9 == 0 else 100 + 11 * i for i in range(72)]
    cases = [
        # uniform partitions + merge kernel (few rows)
        (_case("few_rows", 3, 8, 128, 16, [900, 1500, 700], 71), "partitions"),
        # item list + row table (many rows, a long outlier)
        (_case("item_list", len(long_mixed), 8, 64, 16, long_mixed, 72), "item list"),
        # row table only
        (_case("rows_balanced", len(big_ragged), 8, 128, 16, big_ragged, 73), "row table"),
        # smaller than everything before: partitions again, over the bytes the earlier calls left behind
        (_case("smaller", 2, 4, 96, 16, [300, 17], 74, dtype="bfloat16"), "partitions"),
        # no scratch needed at all
        (_case("no_scratch", 40, 32, 128, 16, [50 + i for i in range(40)], 75), "none"),
        (_case("smaller_again", 1, 2, 64, 8, [129], 76, dtype="float32"), "partitions"),
    ]
    for case, form in cases:
        got, want, ref, nbytes = _both(pkg, case)
        assert (nbytes > 0) == (form != "none"), (case.name, nbytes)
        assert nbytes <= pkg._capi.DECODE_STATIC_SCRATCH_BYTES
        assert not bool(torch.isnan(got.float()).any()), f"{case.name}: rows left unwritten"
        assert torch.equal(got, want), f"{case.name} ({form}): static-scratch result differs from the _ws form"
        atol, rtol = helpers.TIGHT[case.dtype]
        if case.dtype == "bfloat16":
            atol = helpers.decode_atol(datagen.make_decode_inputs(case), case.dtype)
        helpers.assert_close(helpers.out_to_f32(got, case.dtype), ref, atol, 2 * rtol, f"{case.name} vs oracle")


def test_static_scratch_decode_rejects_launches_beyond_its_size(pkg):
    import torch
    dev = "cuda:0"
    S, H, D, B, mc = 1024, 32, 128, 16, 4096
    assert pkg._capi.decode_workspace_bytes(S, H, D, B, mc) > pkg._capi.DECODE_STATIC_SCRATCH_BYTES
    # the sizes are checked before any pointer is used: small dummy tensors are enough
    q = torch.zeros(S, H, D, dtype=torch.float16, device=dev)
    out = torch.full((S, H, D), 5.0, dtype=torch.float16, device=dev)
    kc = torch.zeros(1, H, D // 8, B, 8, dtype=torch.float16, device=dev)
    vc = torch.zeros(1, H, D, B, dtype=torch.float16, device=dev)
    bt = torch.zeros(S, mc // B, dtype=torch.int32, device=dev)
    cl = torch.ones(S, dtype=torch.int32, device=dev)
    with pytest.raises(pkg._capi.PagedAttnError, match="static scratch"):
        pkg._capi.single_query_cached_kv_attention_static(out, q, kc, vc, 0.1, bt, cl, B, mc)
    torch.cuda.synchronize()
    assert bool((out == 5.0).all())
    # unsupported shapes are reported as by the other entry points
    with pytest.raises(pkg._capi.PagedAttnError, match="head size"):
        pkg._capi.single_query_cached_kv_attention_static(out[:, :, :32], q[:, :, :32], kc, vc, 0.1, bt, cl, B, mc)
