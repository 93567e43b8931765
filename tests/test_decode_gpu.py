"""GPU parity tests of decode attention, driven through the C ABI
(libpaged_attn_mi355x.so via ctypes) and through the `attention_ops` module."""
import numpy as np
import pytest

from oracle import datagen, pa_oracle
from tests import helpers

pytestmark = pytest.mark.gpu

DECODE_META, DECODE_DATA = helpers.load_decode_golden()


def _run_capi(pkg, inp, case, device="cuda:0", split=False):
    """split=False: pa_single_query_cached_kv_attention (one workgroup per
    row).  split=True: the _planned entry point (plan made by this call) with
    the workspace it asks for, so small batches take the partitioned kernel +
    merge kernel, big ones with long contexts the item list, and batches of 16
    units of sequences and more the balanced row order.  split="module": the
    attention_ops function with the reference's own signature
    (csrc/attention.cpp:3-19), what the unchanged reference caller invokes."""
    import torch
    qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, device)
    out = torch.full((case.num_seqs, case.num_heads, case.head_size), float("nan"), dtype=qkv.dtype, device=device)
    if split == "module":
        pkg.attention_ops.single_query_cached_kv_attention(
            out, query, kc, vc, inp["scale"], bt, cl, case.block_size, inp["max_context_len"])
    elif split:
        nbytes = pkg._capi.decode_workspace_bytes(case.num_seqs, case.num_heads, case.head_size,
                                                  case.block_size, inp["max_context_len"])
        ws = torch.full((max(nbytes, 4) // 4,), float("nan"), dtype=torch.float32, device=device)
        pkg._capi.single_query_cached_kv_attention_planned(
            out, query, kc, vc, inp["scale"], bt, cl, case.block_size, inp["max_context_len"],
            ws if nbytes else None, False)
    else:
        pkg._capi.single_query_cached_kv_attention(
            out, query, kc, vc, inp["scale"], bt, cl, case.block_size, inp["max_context_len"])
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("split", [False, True, "module"], ids=["rows", "partitioned", "module"])
@pytest.mark.parametrize("name", sorted(DECODE_META))
def test_decode_matches_reference_fixture(pkg, name, split):
    """Every golden case: dtype x block size x head size grid at two input
    scales, edge-case context lengths, and the reference test's own sizes
    (37 tokens, 3 heads, 1024 blocks, context <= 4096) — through the C ABI's two
    forms and through the module function with the reference's signature."""
    entry = DECODE_META[name]
    case = helpers.decode_case_from_meta(entry)
    inp = datagen.make_decode_inputs(case)
    assert datagen.checksum(inp) == entry["inputs_sha256"]
    got = helpers.out_to_f32(_run_capi(pkg, inp, case, split=split), case.dtype)
    atol, rtol = helpers.decode_atol(inp, case.dtype), helpers.TIGHT[case.dtype][1]
    helpers.assert_close(got, DECODE_DATA[name + "/ref_f32"], atol, rtol, f"{name} vs reference fp32")
    if case.dist == "uniform1e-3":
        want = datagen.storage_to_float32(DECODE_DATA[name + "/ref_dtype"], case.dtype)
        helpers.assert_close(got, want, helpers.REF_ATOL, helpers.REF_RTOL, f"{name} vs reference in {case.dtype}")


def test_partition_plan_splits_small_batches_only(pkg):
    c = pkg._capi
    # 256 x 32 rows fill the chip on their own: no partitions, only the row table of the
    # balanced order (8 XCDs x 32 sequences x 8 bytes, and a stamp)
    assert c.decode_workspace_bytes(256, 32, 128, 16, 1024) == (8 * 32 + 1) * 8
    # too few sequences to balance, contexts too short to split
    assert c.decode_workspace_bytes(40, 32, 128, 16, 256) == 0
    # 1280 rows with contexts of several 256-token partitions: partial results for up to 4 partitions per row
    n = c.decode_workspace_bytes(40, 32, 128, 16, 1024)
    assert n >= 40 * 32 * 4 * (128 + 2) * 4
    # one sequence: 32 rows -> split into partitions of at least 64 tokens / 4 blocks
    n = c.decode_workspace_bytes(1, 32, 128, 16, 1024)
    assert n > 0 and n % ((128 + 2) * 4 * 32) == 0
    parts = n // ((128 + 2) * 4 * 32)
    assert 2 <= parts <= 16
    # a context that is a single partition long is never split
    assert c.decode_workspace_bytes(1, 1, 64, 16, 64) == 0
    assert c.decode_workspace_bytes(0, 32, 128, 16, 1024) == 0


def test_partitioned_decode_context_longer_than_announced(pkg):
    """The partition plan is made from max_context_len; a longer context_len
    (caller error in the reference too, where it overruns shared memory) must
    still be attended in full: the last partition runs to the end."""
    import torch
    case = datagen.DecodeCase(name="over", dtype="float16", num_seqs=2, num_heads=2, head_size=64,
                              block_size=16, num_blocks=64, context_lens=[300, 40], dist="normal",
                              tables="permutation", seed=12)
    inp = datagen.make_decode_inputs(case)
    ref = pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], case.dtype)
    inp["max_context_len"] = 200  # understated
    got = helpers.out_to_f32(_run_capi(pkg, inp, case, split=True), case.dtype)
    helpers.assert_close(got, datagen.storage_to_float32(ref, case.dtype), 2e-5, 1e-3, "understated max_context_len")


@pytest.mark.parametrize("split", [False, True], ids=["rows", "partitioned"])
@pytest.mark.parametrize("dtype", datagen.DTYPES)
def test_decode_matches_oracle_ragged_batch(pkg, dtype, split):
    """A larger ragged batch against the C oracle, including the scale
    argument (the reference's Python oracle hard-codes 1/sqrt(d))."""
    case = datagen.DecodeCase(name="ragged", dtype=dtype, num_seqs=33, num_heads=5, head_size=128,
                              block_size=16, num_blocks=700, max_context=600, dist="normal",
                              tables="permutation", scale=0.11, seed=21)
    inp = datagen.make_decode_inputs(case)
    got = helpers.out_to_f32(_run_capi(pkg, inp, case, split=split), dtype)
    ref = pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], dtype)
    atol, rtol = helpers.decode_atol(inp, dtype), helpers.TIGHT[dtype][1]
    helpers.assert_close(got, datagen.storage_to_float32(ref, dtype), atol, 2 * rtol, f"ragged {dtype}")


def test_decode_via_attention_ops_module_like_the_reference_test(pkg):
    """tests/kernels/test_attention.py:159-224 with its input distribution
    and tolerance, through the pybind11 module."""
    import torch
    for dtype in datagen.DTYPES:
        for B in (8, 16, 32):
            D = {8: 80, 16: 128, 32: 96}[B]
            case = datagen.DecodeCase(name="mod", dtype=dtype, num_seqs=37, num_heads=3, head_size=D,
                                      block_size=B, num_blocks=256, max_context=1024,
                                      dist="uniform1e-3", seed=33 + B)
            inp = datagen.make_decode_inputs(case)
            qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, dtype, "cuda:0")
            output = torch.empty(case.num_seqs, case.num_heads, D, dtype=qkv.dtype, device="cuda:0")
            pkg.attention_ops.single_query_cached_kv_attention(
                output, query, kc, vc, inp["scale"], bt, cl, B, inp["max_context_len"])
            ref = pa_oracle.single_query_cached_kv_attention(
                inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
                inp["context_lens"], dtype)
            ref_t = datagen.storage_to_torch(ref, dtype).to("cuda:0")
            assert torch.allclose(output, ref_t, atol=1e-3, rtol=1e-5), (dtype, B, D)


def test_decode_masked_tail_ignores_garbage_in_the_cache(pkg):
    """Slots past context_len in the last block may hold anything (the cache
    is torch.empty in vllm/worker/cache_engine.py:70-79), including NaN."""
    import torch
    case = datagen.DecodeCase(name="nan", dtype="float16", num_seqs=4, num_heads=2, head_size=128,
                              block_size=16, num_blocks=32, context_lens=[3, 17, 31, 16], dist="normal",
                              tables="permutation", seed=8)
    inp = datagen.make_decode_inputs(case)
    ref = pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], case.dtype)
    kc, vc = inp["key_cache"], inp["value_cache"]
    nan = np.float16(np.nan)
    for s, c in enumerate(inp["context_lens"]):
        last = inp["block_tables"][s, (c - 1) // 16]
        first_bad = c % 16
        if first_bad:
            kc[last, :, :, first_bad:, :] = nan
            vc[last, :, :, first_bad:] = nan
    got = helpers.out_to_f32(_run_capi(pkg, inp, case), case.dtype)
    helpers.assert_close(got, datagen.storage_to_float32(ref, case.dtype), 2e-5, 1e-3, "NaN tail")


def test_decode_zero_context_and_empty_batch(pkg):
    import torch
    case = datagen.DecodeCase(name="zero", dtype="float16", num_seqs=3, num_heads=2, head_size=64,
                              block_size=16, num_blocks=8, context_lens=[0, 5, 0], dist="normal", seed=2)
    inp = datagen.make_decode_inputs(case)
    got = helpers.out_to_f32(_run_capi(pkg, inp, case), case.dtype)
    assert np.all(got[0] == 0) and np.all(got[2] == 0)
    ref = pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], case.dtype)
    helpers.assert_close(got[1], datagen.storage_to_float32(ref, case.dtype)[1], 2e-5, 1e-3, "ctx 5")
    # empty batch: nothing is launched, nothing is touched
    q = torch.empty(0, 2, 64, dtype=torch.float16, device="cuda:0")
    o = torch.empty(0, 2, 64, dtype=torch.float16, device="cuda:0")
    kc = torch.zeros(8, 2, 8, 16, 8, dtype=torch.float16, device="cuda:0")
    vc = torch.zeros(8, 2, 64, 16, dtype=torch.float16, device="cuda:0")
    bt = torch.zeros(0, 1, dtype=torch.int32, device="cuda:0")
    cl = torch.zeros(0, dtype=torch.int32, device="cuda:0")
    pkg.attention_ops.single_query_cached_kv_attention(o, q, kc, vc, 0.125, bt, cl, 16, 0)
    torch.cuda.synchronize()


def test_decode_unaligned_query_rows(pkg):
    """A query view whose rows are not 16-byte aligned takes the element-wise
    q load path."""
    import torch
    case = datagen.DecodeCase(name="unal", dtype="float16", num_seqs=5, num_heads=3, head_size=80,
                              block_size=8, num_blocks=64, max_context=90, dist="normal", seed=4)
    inp = datagen.make_decode_inputs(case)
    qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, "cuda:0")
    S, H, D = query.shape
    buf = torch.zeros(S * (H * D + 3) + 1, dtype=query.dtype, device="cuda:0")
    q_odd = buf[1:].as_strided((S, H, D), (H * D + 3, D, 1))
    q_odd.copy_(query)
    out = torch.empty(S, H, D, dtype=query.dtype, device="cuda:0")
    pkg._capi.single_query_cached_kv_attention(out, q_odd, kc, vc, inp["scale"], bt, cl, case.block_size,
                                               inp["max_context_len"])
    torch.cuda.synchronize()
    ref = pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], case.dtype)
    helpers.assert_close(helpers.out_to_f32(out, case.dtype), datagen.storage_to_float32(ref, case.dtype),
                         2e-5, 1e-3, "unaligned q")


def test_decode_rejects_what_the_reference_rejects(pkg):
    """Unsupported head size / block size / dtype raise, as TORCH_CHECK does in
    csrc/attention/attention_kernels.cu:398,446,467."""
    import torch
    dev = "cuda:0"

    def call(D, B, dtype):
        x = 16 // torch.tensor([], dtype=dtype).element_size()
        q = torch.zeros(2, 2, D, dtype=dtype, device=dev)
        o = torch.zeros(2, 2, D, dtype=dtype, device=dev)
        kc = torch.zeros(4, 2, max(D // x, 1), B, x, dtype=dtype, device=dev)
        vc = torch.zeros(4, 2, D, B, dtype=dtype, device=dev)
        bt = torch.zeros(2, 1, dtype=torch.int32, device=dev)
        cl = torch.ones(2, dtype=torch.int32, device=dev)
        pkg.attention_ops.single_query_cached_kv_attention(o, q, kc, vc, 1.0, bt, cl, B, 1)

    with pytest.raises(RuntimeError, match="head size"):
        call(32, 16, torch.float16)
    with pytest.raises(RuntimeError, match="block size"):
        call(64, 64, torch.float16)
    with pytest.raises(RuntimeError, match="data type"):
        call(64, 16, torch.float64)
    call(64, 16, torch.float16)
    torch.cuda.synchronize()


def test_decode_north_star_size_properties(pkg):
    """bs 256 / 32 heads / head 128 / ctx 1024 / block 16 fp16, distinct
    physical blocks.  Too big for the oracle in full, so check size-independent
    properties and a sampled subset of rows:
      * V == 1 everywhere  => every output element is sum(p) = 1 / (1 + 1e-6 / sum e)
      * a sample of (sequence, head) rows equals the oracle on those rows."""
    import torch
    dev = "cuda:0"
    S, H, D, B, CTX = 256, 32, 128, 16, 1024
    nb = S * CTX // B
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    qkv = torch.randn(S, 3, H, D, dtype=torch.float16, device=dev, generator=g) * 0.5
    query = qkv[:, 0]
    kc = torch.randn(nb, H, D // 8, B, 8, dtype=torch.float16, device=dev, generator=g) * 0.5
    vc = torch.ones(nb, H, D, B, dtype=torch.float16, device=dev)
    perm = torch.randperm(nb, device=dev, generator=g).to(torch.int32).reshape(S, CTX // B)
    ctx_list = [CTX] * S
    ctx_list[1], ctx_list[2], ctx_list[3] = 1, CTX - 15, 17
    cl = torch.tensor(ctx_list, dtype=torch.int32, device=dev)
    out = torch.empty(S, H, D, dtype=torch.float16, device=dev)
    scale = 1.0 / D ** 0.5
    pkg._capi.single_query_cached_kv_attention(out, query, kc, vc, scale, perm, cl, B, CTX)
    torch.cuda.synchronize()
    assert torch.allclose(out.float(), torch.ones_like(out, dtype=torch.float32), atol=1e-3), \
        "softmax weights do not sum to one"
    # sampled rows against the oracle, with random V
    vc.copy_(torch.randn(nb, H, D, B, dtype=torch.float16, device=dev, generator=g) * 0.5)
    pkg._capi.single_query_cached_kv_attention(out, query, kc, vc, scale, perm, cl, B, CTX)
    torch.cuda.synchronize()
    rows = [0, 1, 2, 3, 100, 255]
    heads = [0, 13, 31]
    for s in rows:
        blocks = perm[s].long()
        # compact this sequence's blocks and the sampled heads into a small problem
        kcs = kc[blocks][:, heads].contiguous().cpu().numpy()
        vcs = vc[blocks][:, heads].contiguous().cpu().numpy()
        qs = query[s:s + 1][:, heads].contiguous().cpu().numpy()
        tbl = np.arange(len(blocks), dtype=np.int32)[None, :]
        ref = pa_oracle.single_query_cached_kv_attention(
            qs, kcs, vcs, scale, tbl, np.array([ctx_list[s]], dtype=np.int32), "float16")
        got = out[s][heads].float().cpu().numpy()
        helpers.assert_close(got, ref.astype(np.float32)[0], 2e-5, 1.5e-3, f"north-star row {s}")


@pytest.mark.parametrize("num_heads", [1, 8, 32])
def test_decode_work_order_covers_every_row_for_any_batch_size(pkg, num_heads):
    """The kernel walks (sequence, head) rows in a permuted order: by batch
    position (groups of sequences dealt round-robin to the XCDs, a partial
    group last), or, given a workspace, by the plan kernel's row table
    (sequences placed by length).  Batch sizes on both sides of the group
    size, of 8 groups, of 16 units and of the 1024-sequence limit, context
    lengths sorted: every row must be written, and match the oracle."""
    import torch
    G = max(1, min(16, 128 // num_heads))
    U = min(G, 4)
    sizes = sorted({1, G - 1, G, G + 1, 7 * G, 8 * G - 1, 8 * G, 8 * G + 1, 9 * G + 3, 17 * G,
                    15 * U + 1, 16 * U, 16 * U + 1, 24 * U + 2, 257, 1024, 1025} - {0})
    for S in sizes:
        if S * num_heads > 12000:
            continue  # keeps the oracle fast; the 32-head case stops at 257 sequences
        ctx = sorted((1 + (37 * i) % 90 for i in range(S)), reverse=True)
        if S % 2:
            ctx = ctx[::-1]
        case = datagen.DecodeCase(name=f"order{S}", dtype="float16", num_seqs=S, num_heads=num_heads, head_size=64,
                                  block_size=16, num_blocks=6 * S + 8, context_lens=ctx, dist="normal",
                                  tables="permutation", seed=300 + S)
        inp = datagen.make_decode_inputs(case)
        ref = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
            inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
            inp["context_lens"], case.dtype), case.dtype)
        # without a workspace: batch order; with one: from 16 units of sequences on, up to 1024
        # sequences, the order of the plan kernel's row table
        for split in (False, True):
            out = _run_capi(pkg, inp, case, split=split)
            assert not bool(torch.isnan(out.float()).any()), f"S={S} split={split}: rows left unwritten"
            helpers.assert_close(helpers.out_to_f32(out, case.dtype), ref, 2e-5, 1e-3, f"S={S} H={num_heads} split={split}")


@pytest.mark.parametrize("reps", [4, 6], ids=["batch_order", "balanced_order"])
@pytest.mark.parametrize("dtype", ["float16", "float32"])
def test_decode_item_list_mode_on_a_long_mixed_batch(pkg, dtype, reps):
    """More than 256 rows and a longest context beyond 2048 tokens: the library
    lists the non-empty (sequence, partition) pairs on the device and a
    fixed-size grid draws from the list.  Mixed lengths on both sides of the
    partition edges, empty contexts, one long outlier; also with an
    understated max_context_len (the last partition must run to the end), and with the
    sum of the context lengths given (fewer workgroup ids reserved for partitions)."""
    import torch
    ctx = [5000, 100, 2049, 1024, 1025, 0, 1, 16, 2048, 3000, 1023, 77] * reps + [4097]
    S, H, D, B = len(ctx), 8, 64, 16  # 16 units of 4 sequences and more take the balanced order
    case = datagen.DecodeCase(name="list", dtype=dtype, num_seqs=S, num_heads=H, head_size=D, block_size=B,
                              num_blocks=sum((c + B - 1) // B for c in ctx) + 5, context_lens=ctx, dist="normal",
                              tables="permutation", seed=77)
    inp = datagen.make_decode_inputs(case)
    assert S * H > 256 and inp["max_context_len"] > 2048
    assert pkg._capi.decode_workspace_bytes(S, H, D, B, inp["max_context_len"]) > 0
    ref = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], dtype), dtype)
    atol, rtol = helpers.decode_atol(inp, dtype), helpers.TIGHT[dtype][1]
    for announced in (inp["max_context_len"], 3000):
        inp["max_context_len"] = announced
        for _ in range(2):  # twice on one workspace: the plan kernel rewinds the counters
            got = helpers.out_to_f32(_run_capi(pkg, inp, case, split=True), dtype)
            helpers.assert_close(got, ref, atol, 2 * rtol, f"item list, announced {announced}")
    # with the sum of the context lengths: workgroup ids are reserved for as many partitions as
    # that sum allows, not for the worst case; a sum that is too small is caught on the device
    inp["max_context_len"] = max(ctx)
    qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, dtype, "cuda:0")
    nbytes = pkg._capi.decode_workspace_bytes(S, H, D, B, max(ctx))
    ws = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device="cuda:0")
    for total, good in ((sum(ctx), True), (sum(ctx) // 8, False)):
        for ready in (False, True):
            out = torch.full((S, H, D), 2.0, dtype=qkv.dtype, device="cuda:0")
            pkg._capi.single_query_cached_kv_attention_planned(out, query, kc, vc, inp["scale"], bt, cl, B, max(ctx), ws,
                                                               ready, total)
            torch.cuda.synchronize()
            if good:
                helpers.assert_close(helpers.out_to_f32(out, dtype), ref, atol, 2 * rtol, f"total given, ready={ready}")
            else:
                assert bool(torch.isnan(out.float()).all()), "an understated total_context_len passed for a result"


@pytest.mark.parametrize("seed", range(24))
def test_decode_randomized_shapes_against_oracle(pkg, seed):
    """Differential test over random shapes: batch size, head count, head size,
    block size, dtype, context-length mixture (empty, short, around the
    1024-token partition edge, long), padded tables, with and without the
    workspace — so that every launch form (rows, uniform partitions, item list)
    and both workgroup widths meet odd sizes."""
    rng = np.random.default_rng(1000 + seed)
    dtype = ["float16", "bfloat16", "float32"][seed % 3]
    D = int(rng.choice([64, 80, 96, 128]))
    B = int(rng.choice([8, 16, 32]))
    H = int(rng.choice([1, 2, 3, 5, 8]))
    S = int(rng.choice([1, 2, 7, 33, 70, 130, 300]))
    kinds = rng.choice(5, size=S, p=[0.05, 0.45, 0.2, 0.2, 0.1])
    ctx = []
    for k in kinds:
        if k == 0:
            ctx.append(0)
        elif k == 1:
            ctx.append(int(rng.integers(1, 200)))
        elif k == 2:
            ctx.append(int(rng.integers(1000, 1050)))
        elif k == 3:
            ctx.append(int(rng.integers(200, 1000)))
        else:
            ctx.append(int(rng.integers(1050, 2600)))
    if S * H * sum(ctx) * D > 4e8:  # keep the oracle fast
        ctx = [c // 4 for c in ctx]
    case = datagen.DecodeCase(name=f"fuzz{seed}", dtype=dtype, num_seqs=S, num_heads=H, head_size=D, block_size=B,
                              num_blocks=sum((c + B - 1) // B for c in ctx) + 3, context_lens=ctx, dist="normal",
                              tables="permutation", table_pad=int(rng.integers(0, 4)), scale=float(rng.uniform(0.05, 0.2)),
                              seed=2000 + seed)
    inp = datagen.make_decode_inputs(case)
    ref = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], dtype), dtype)
    atol, rtol = helpers.decode_atol(inp, dtype), helpers.TIGHT[dtype][1]
    for split in (False, True):
        got = helpers.out_to_f32(_run_capi(pkg, inp, case, split=split), dtype)
        helpers.assert_close(got, ref, atol, 2 * rtol, f"seed {seed} S={S} H={H} D={D} B={B} {dtype} split={split}")


def _list_layout_bytes(S, H, D, B, max_ctx, parts_cap, part_tokens, need):
    """Bytes of the item-list workspace layout with at most parts_cap partitions of at least
    part_tokens tokens, given `need` = the bytes of the layout with 16 x 256 (what
    pa_decode_workspace_bytes reports when that is the larger one): the two differ in the
    partial results (rows x parts x (D + 2) floats) and the item list (S x parts words)."""
    nw = 8 if D <= 64 else 4
    mb = (max_ctx + B - 1) // B

    def parts(cap, tokens):
        bpp = max(tokens // B, (mb + cap - 1) // cap)
        bpp = (bpp + nw - 1) // nw * nw
        return (mb + bpp - 1) // bpp

    fine, coarse = parts(16, 256), parts(parts_cap, part_tokens)
    return need - (fine - coarse) * (S * H * (D + 2) * 4 + S * 4)


def test_decode_item_list_plan_splits_only_rows_that_would_hold_up_the_launch(pkg):
    """Item-list mode decides on the device whether partitions pay: they do
    when the longest sequence exceeds half the average work per resident
    workgroup slot (blocks x heads / slots; 512 slots for head size 64) —
    unless all rows are resident at once and the longest is within a quarter of
    the mean length, when nothing needs evening out.  A batch of 160
    equal-length sequences keeps one workgroup per row (no listed pairs,
    partition length 'infinite'); the same workspace then serves a mixed batch,
    which is split, a batch of 40 equal-length sequences (320 rows, all
    resident: not split) and 40 sequences with one long one among them (split).
    The plan sits in the last three words of the workspace.  All results must
    match the oracle."""
    import torch
    H, D, B = 8, 64, 16
    uniform = [1400 + (13 * i) % 120 for i in range(160)]
    mixed = [3000 if i % 17 == 0 else 150 + i for i in range(160)]
    few = uniform[:40]
    few_mixed = [2900 if i == 7 else c for i, c in enumerate(few)]
    nbytes = max(pkg._capi.decode_workspace_bytes(len(c), H, D, B, max(c)) for c in (uniform, mixed, few, few_mixed))
    ws = torch.zeros(nbytes // 4 + 2, dtype=torch.float32, device="cuda:0")
    for name, ctx, split_expected in (("uniform", uniform, False), ("mixed", mixed, True), ("few rows", few, False),
                                      ("few rows, one long", few_mixed, True), ("uniform again", uniform, False)):
        case = datagen.DecodeCase(name="plan", dtype="float16", num_seqs=len(ctx), num_heads=H, head_size=D, block_size=B,
                                  num_blocks=sum((c + B - 1) // B for c in ctx) + 5, context_lens=ctx, dist="normal",
                                  tables="permutation", seed=91)
        inp = datagen.make_decode_inputs(case)
        qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, "cuda:0")
        out = torch.full((len(ctx), H, D), float("nan"), dtype=qkv.dtype, device="cuda:0")
        need = pkg._capi.decode_workspace_bytes(len(ctx), H, D, B, inp["max_context_len"])
        assert need % 4 == 0
        view = ws[:need // 4]
        pkg._capi.single_query_cached_kv_attention_ws(out, query, kc, vc, inp["scale"], bt, cl, B,
                                                      inp["max_context_len"], view)
        torch.cuda.synchronize()
        # a call without the sum of the context lengths plans at most 4 partitions of >= 1024 tokens
        used = _list_layout_bytes(len(ctx), H, D, B, inp["max_context_len"], 4, 1024, need)
        plan = view.view(torch.int32)[used // 4 - 3:used // 4].tolist()
        if split_expected:
            bpp = plan[2]
            assert bpp * B >= 1024 and bpp < (1 << 24), name
            most = ((max(ctx) + B - 1) // B + bpp - 1) // bpp  # partitions of the longest context the plan allows for
            assert 2 <= most <= 4, name
            assert plan[1] == sum(min((c + bpp * B - 1) // (bpp * B), most) - 1 for c in ctx), name
            assert plan[1] > 0, name
        else:
            assert plan[1] == 0 and plan[2] == 1 << 24, name
        ref = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
            inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
            inp["context_lens"], case.dtype), case.dtype)
        atol, rtol = helpers.TIGHT[case.dtype]
        helpers.assert_close(helpers.out_to_f32(out, case.dtype), ref, atol, 2 * rtol, name)


@pytest.mark.parametrize("longest", [300, 2600], ids=["rows", "item_list"])
@pytest.mark.parametrize("entry", ["planned", "ws", "module"])
def test_decode_plan_shared_by_several_calls(pkg, longest, entry):
    """One workspace, one plan, several calls with different queries and caches
    (the layers of a step): plan_ready False, then True.  The workspace starts
    out as NaN bytes.  Also the per-call forms (_ws entry point, attention_ops
    module) on the same inputs."""
    import torch
    rng = np.random.default_rng(5)
    S, H, D, B = 100, 8, 64, 16
    ctx = [int(c) for c in rng.integers(1, longest + 1, size=S)]
    ctx[17] = 0
    ctx[3] = longest
    nbytes = pkg._capi.decode_workspace_bytes(S, H, D, B, longest)
    assert nbytes > 0
    ws = torch.full((nbytes // 4 + 1,), float("nan"), dtype=torch.float32, device="cuda:0")
    for layer in range(3):
        case = datagen.DecodeCase(name="shared", dtype="float16", num_seqs=S, num_heads=H, head_size=D, block_size=B,
                                  num_blocks=sum((c + B - 1) // B for c in ctx) + 5, context_lens=ctx, dist="normal",
                                  tables="permutation", seed=500 + layer)
        inp = datagen.make_decode_inputs(case)
        qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, "cuda:0")
        out = torch.full((S, H, D), float("nan"), dtype=qkv.dtype, device="cuda:0")
        if entry == "planned":
            pkg._capi.single_query_cached_kv_attention_planned(out, query, kc, vc, inp["scale"], bt, cl, B, longest, ws,
                                                               layer > 0)
        elif entry == "ws":
            pkg._capi.single_query_cached_kv_attention_ws(out, query, kc, vc, inp["scale"], bt, cl, B, longest, ws)
        else:
            pkg.attention_ops.single_query_cached_kv_attention(out, query, kc, vc, inp["scale"], bt, cl, B, longest)
        torch.cuda.synchronize()
        ref = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
            inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
            inp["context_lens"], case.dtype), case.dtype)
        atol, rtol = helpers.TIGHT[case.dtype]
        helpers.assert_close(helpers.out_to_f32(out, case.dtype), ref, atol, 2 * rtol, f"{entry} layer {layer}")


def test_decode_total_context_hint_leaves_the_item_list_out(pkg):
    """With the sum of the context lengths known on the host, a big batch whose
    longest context cannot hold up the launch is run as plain rows (in the
    balanced order): no item list is planned — the counters at the end of the
    workspace stay untouched — and the result is the oracle's.  Without the
    hint the same call plans the list on the device (and lists nothing)."""
    import torch
    S, H, D, B = 300, 8, 64, 16
    ctx = [1100 + (37 * i) % 200 for i in range(S)]
    case = datagen.DecodeCase(name="hint", dtype="float16", num_seqs=S, num_heads=H, head_size=D, block_size=B,
                              num_blocks=sum((c + B - 1) // B for c in ctx) + 5, context_lens=ctx, dist="normal",
                              tables="permutation", seed=93)
    inp = datagen.make_decode_inputs(case)
    qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, "cuda:0")
    ref = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], case.dtype), case.dtype)
    nbytes = pkg._capi.decode_workspace_bytes(S, H, D, B, max(ctx))
    assert nbytes > 8 * 8 * S  # more than a row table: the list form is planned for
    atol, rtol = helpers.TIGHT[case.dtype]
    for total, counters_written in ((sum(ctx), False), (-1, True)):
        ws = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device="cuda:0")
        for ready in (False, True):
            out = torch.full((S, H, D), float("nan"), dtype=qkv.dtype, device="cuda:0")
            pkg._capi.single_query_cached_kv_attention_planned(out, query, kc, vc, inp["scale"], bt, cl, B, max(ctx), ws,
                                                               ready, total)
            torch.cuda.synchronize()
            helpers.assert_close(helpers.out_to_f32(out, case.dtype), ref, atol, 2 * rtol, f"total={total} ready={ready}")
        # without the sum the plan is the 4 x 1024-token one, whose layout ends before the workspace does
        end = _list_layout_bytes(S, H, D, B, max(ctx), 4, 1024, nbytes) // 4
        assert bool(torch.isnan(ws[-3:]).all())
        assert bool(torch.isnan(ws[end - 3:end]).all()) != counters_written
        if counters_written:
            assert ws.view(torch.int32)[end - 3:end].tolist()[1:] == [0, 1 << 24]


def test_decode_equal_long_contexts_take_two_uniform_partitions(pkg):
    """A batch between one and two residencies of rows (40 sequences x 32 heads against the 768 workgroups
    the GPU holds) whose contexts are all (nearly) as long as the longest: with the sum of the context lengths
    known, the launch takes two uniform partitions per row instead of the item list — no plan kernel, so the
    tail of the workspace (where the list's items and counters would sit) stays untouched — and the result is
    the oracle's.  A batch half of whose sequences are short keeps the list; both forms agree with the oracle."""
    import torch
    S, H, D, B = 40, 32, 128, 16
    for name, ctx in (("equal", [2048 - (i % 3) * 16 for i in range(S)]), ("half_short", [2048] * (S // 2) + [300] * (S // 2))):
        case = datagen.DecodeCase(name=name, dtype="float16", num_seqs=S, num_heads=H, head_size=D, block_size=B,
                                  num_blocks=sum((c + B - 1) // B for c in ctx) + 5, context_lens=ctx, dist="normal",
                                  tables="permutation", seed=97)
        inp = datagen.make_decode_inputs(case)
        qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, "cuda:0")
        ref = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
            inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
            inp["context_lens"], case.dtype), case.dtype)
        nbytes = pkg._capi.decode_workspace_bytes(S, H, D, B, max(ctx))
        atol, rtol = helpers.TIGHT[case.dtype]
        ws = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device="cuda:0")
        for ready in (False, True):
            out = torch.full((S, H, D), float("nan"), dtype=qkv.dtype, device="cuda:0")
            pkg._capi.single_query_cached_kv_attention_planned(out, query, kc, vc, inp["scale"], bt, cl, B, max(ctx), ws,
                                                               ready, sum(ctx))
            torch.cuda.synchronize()
            helpers.assert_close(helpers.out_to_f32(out, case.dtype), ref, atol, 2 * rtol, f"{name} ready={ready}")
        # two partitions per row: accumulators and (max, sum) pairs, nothing behind them
        used = S * H * 2 * (D + 2)
        assert not bool(torch.isnan(ws[:used]).any()) or name != "equal"
        assert bool(torch.isnan(ws[used:]).all()) == (name == "equal")


def test_decode_step_replays_from_a_hip_graph(pkg):
    """A decode step of three layers (reshape_and_cache, then attention with the
    plan shared by the layers; and the module function with the reference's
    signature, which takes its scratch from torch's allocator) captured into a
    HIP graph and replayed with new queries: same results as eager launches."""
    import torch
    dev = "cuda:0"
    S, H, D, B, L = 70, 8, 64, 16, 3
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    ctx = torch.randint(1, 1500, (S,), generator=g, device=dev, dtype=torch.int32)
    ctx[5] = 2300
    max_ctx = int(ctx.max())
    bps = (max_ctx + B - 1) // B
    NB = S * bps
    bt = torch.randperm(NB, generator=g, device=dev).to(torch.int32).reshape(S, bps)
    slots = (bt[torch.arange(S, device=dev), ((ctx - 1) // B).long()].long() * B + ((ctx - 1) % B).long()).to(torch.int32)
    kcs = [torch.randn(NB, H, D // 8, B, 8, dtype=torch.float16, device=dev, generator=g) for _ in range(L)]
    vcs = [torch.randn(NB, H, D, B, dtype=torch.float16, device=dev, generator=g) for _ in range(L)]
    qkv = torch.randn(S, 3, H, D, dtype=torch.float16, device=dev, generator=g)
    outs = [torch.zeros(S, H, D, dtype=torch.float16, device=dev) for _ in range(2 * L)]
    ws = torch.empty(pkg.attention_ops.decode_workspace_bytes(S, H, D, B, max_ctx), dtype=torch.uint8, device=dev)
    total = int(ctx.sum())

    def step():
        for l in range(L):
            pkg.cache_ops.reshape_and_cache(qkv[:, 1], qkv[:, 2], kcs[l], vcs[l], slots)
            pkg.attention_ops.single_query_cached_kv_attention_planned(
                outs[l], qkv[:, 0], kcs[l], vcs[l], D ** -0.5, bt, ctx, B, max_ctx, ws, l > 0, total)
            pkg.attention_ops.single_query_cached_kv_attention(
                outs[L + l], qkv[:, 0], kcs[l], vcs[l], D ** -0.5, bt, ctx, B, max_ctx)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()  # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for trial in range(2):
        qkv.copy_(torch.randn(S, 3, H, D, dtype=torch.float16, device=dev, generator=g))
        graph.replay()
        torch.cuda.synchronize()
        replayed = [o.clone() for o in outs]
        caches = [c.clone() for c in kcs + vcs]
        step()
        torch.cuda.synchronize()
        for a, b in zip(replayed, outs):
            assert torch.equal(a, b), trial
        for a, b in zip(caches, kcs + vcs):
            assert torch.equal(a, b), trial
        for l in range(L):  # the two entry points agree up to the summation order of their launch forms
            assert torch.allclose(outs[l].float(), outs[L + l].float(), atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("longest", [300, 2600], ids=["rows", "item_list"])
def test_decode_with_a_workspace_that_holds_no_plan_poisons_the_output(pkg, longest):
    """plan_ready=True on a workspace whose plan was never made (arbitrary
    bytes), or was made for another batch size: every workgroup sees that the
    stamp does not match, computes nothing and fills `out` with NaN, so the
    caller's error cannot pass for a result.  The same workspace then serves a
    proper pair of calls."""
    import torch
    rng = np.random.default_rng(8)
    H, D, B = 8, 64, 16
    results = {}
    ws = None
    for S in (100, 96):
        ctx = [int(c) for c in rng.integers(1, longest + 1, size=S)]
        ctx[3] = longest
        case = datagen.DecodeCase(name="noplan", dtype="float16", num_seqs=S, num_heads=H, head_size=D, block_size=B,
                                  num_blocks=sum((c + B - 1) // B for c in ctx) + 5, context_lens=ctx, dist="normal",
                                  tables="permutation", seed=600 + S)
        inp = datagen.make_decode_inputs(case)
        qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, "cuda:0")
        nbytes = pkg._capi.decode_workspace_bytes(S, H, D, B, longest)
        if ws is None:  # S = 100: arbitrary bytes;  S = 96: the plan of the S = 100 batch
            ws = torch.randint(0, 2 ** 31 - 1, (nbytes // 4 + 64,), dtype=torch.int32, device="cuda:0")
        out = torch.full((S, H, D), 7.0, dtype=qkv.dtype, device="cuda:0")
        pkg._capi.single_query_cached_kv_attention_planned(out, query, kc, vc, inp["scale"], bt, cl, B, longest,
                                                           ws[:nbytes // 4], True)
        torch.cuda.synchronize()
        assert bool(torch.isnan(out.float()).all()), f"S={S}: a call without a valid plan left values in out"
        ref = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
            inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
            inp["context_lens"], case.dtype), case.dtype)
        atol, rtol = helpers.TIGHT[case.dtype]
        for ready in (False, True):
            out.fill_(float("nan"))
            pkg._capi.single_query_cached_kv_attention_planned(out, query, kc, vc, inp["scale"], bt, cl, B, longest,
                                                               ws[:nbytes // 4], ready)
            torch.cuda.synchronize()
            helpers.assert_close(helpers.out_to_f32(out, case.dtype), ref, atol, 2 * rtol, f"S={S} ready={ready}")


def _balanced_slots_per_xcd(S, H):
    """Slots per XCD of the balanced order's row table (the library's plan_balance rule)."""
    group = min(max(128 // H, 1), 16)
    unit = min(4, group)
    group = group // unit * unit
    units = (S + unit - 1) // unit
    assert units >= 16
    return ((units + 7) // 8 * unit + group - 1) // group * group


def test_decode_item_list_with_a_foreign_row_table_poisons_every_row(pkg):
    """Item-list form on top of the balanced order: the list's stamp (counters) is intact but the row
    table's is not (its words were overwritten).  Every output row must come out NaN — also the rows whose
    index is below the number of workgroup ids reserved for listed partitions."""
    import torch
    rng = np.random.default_rng(18)
    S, H, D, B, longest = 100, 8, 64, 16, 2600
    ctx = [int(c) for c in rng.integers(1, longest + 1, size=S)]
    ctx[3] = longest
    case = datagen.DecodeCase(name="stamp", dtype="float16", num_seqs=S, num_heads=H, head_size=D, block_size=B,
                              num_blocks=sum((c + B - 1) // B for c in ctx) + 5, context_lens=ctx, dist="normal",
                              tables="permutation", seed=640)
    inp = datagen.make_decode_inputs(case)
    qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, "cuda:0")
    nbytes = pkg._capi.decode_workspace_bytes(S, H, D, B, longest)
    ws = torch.zeros(nbytes // 4 + 1, dtype=torch.int32, device="cuda:0")
    out = torch.full((S, H, D), 7.0, dtype=qkv.dtype, device="cuda:0")
    args = (out, query, kc, vc, inp["scale"], bt, cl, B, longest, ws[:nbytes // 4])
    pkg._capi.single_query_cached_kv_attention_planned(*args, False)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(out.float()).any())
    at = 8 * _balanced_slots_per_xcd(S, H) * 2  # the stamp sits behind the table's [8][slots] (sequence, length) pairs
    stamp = ws[at:at + 2].tolist()
    assert stamp[0] & 1 and (stamp[0] ^ stamp[1]) & 0xffffffff == 0xffffffff, stamp
    ws[at] ^= 0x10
    out.fill_(7.0)
    pkg._capi.single_query_cached_kv_attention_planned(*args, True)
    torch.cuda.synchronize()
    assert bool(torch.isnan(out.float()).all()), "rows kept stale values although the row table's stamp does not match"


@pytest.mark.parametrize("longest", [300, 2600], ids=["rows", "item_list"])
def test_decode_plan_reused_after_context_lens_changed_poisons_the_changed_rows(pkg, longest):
    """A plan depends on context_lens.  A caller that reuses one (plan_ready=True) after some lengths
    changed — same sizes, other contents — must not get plausible numbers for those sequences: their rows
    come out NaN; the rows of the unchanged sequences are still right (the plan only decides where and in
    how many pieces a row is computed)."""
    import torch
    rng = np.random.default_rng(28)
    S, H, D, B = 100, 8, 64, 16
    ctx = [int(c) for c in rng.integers(1, longest + 1, size=S)]
    ctx[3] = longest
    changed = {5: 17, 40: longest - 3, 41: ctx[41] + 1 if ctx[41] < longest else ctx[41] - 1, 99: 1}
    changed = {k: v for k, v in changed.items() if v != ctx[k]}
    assert len(changed) >= 3
    ctx2 = [changed.get(i, c) for i, c in enumerate(ctx)]
    case = datagen.DecodeCase(name="stale", dtype="float16", num_seqs=S, num_heads=H, head_size=D, block_size=B,
                              num_blocks=S * ((longest + B - 1) // B) + 5, context_lens=ctx2, dist="normal",
                              tables="permutation", seed=650)
    inp = datagen.make_decode_inputs(case)
    qkv, query, kc, vc, bt, cl2 = helpers.decode_inputs_to_device(inp, case.dtype, "cuda:0")
    cl = torch.tensor(ctx, dtype=torch.int32, device="cuda:0")
    nbytes = pkg._capi.decode_workspace_bytes(S, H, D, B, longest)
    ws = torch.zeros(nbytes // 4 + 1, dtype=torch.int32, device="cuda:0")
    out = torch.full((S, H, D), 7.0, dtype=qkv.dtype, device="cuda:0")
    # the plan is made from ctx (the tables are wide enough for either set of lengths) ...
    pkg._capi.single_query_cached_kv_attention_planned(out, query, kc, vc, inp["scale"], bt, cl, B, longest,
                                                       ws[:nbytes // 4], False)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(out.float()).any())
    # ... and reused with ctx2
    out.fill_(7.0)
    pkg._capi.single_query_cached_kv_attention_planned(out, query, kc, vc, inp["scale"], bt, cl2, B, longest,
                                                       ws[:nbytes // 4], True)
    torch.cuda.synchronize()
    got = helpers.out_to_f32(out, case.dtype)
    ref = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], case.dtype), case.dtype)
    same = [i for i in range(S) if i not in changed]
    for i in changed:
        assert np.isnan(got[i]).all(), f"sequence {i}: length changed under a reused plan, row not poisoned"
    atol, rtol = helpers.TIGHT[case.dtype]
    helpers.assert_close(got[same], ref[same], atol, 2 * rtol, "unchanged sequences under a reused plan")


def _storage_inputs_as_float32(inp, dtype):
    """The same numbers as float32 arrays in float32's cache layout (x = 4 instead of 8)."""
    f = lambda a: datagen.storage_to_float32(a, dtype)  # noqa: E731
    kc = f(inp["key_cache"])
    NB, H, DX, B, X = kc.shape
    D = DX * X
    kc = kc.transpose(0, 1, 3, 2, 4).reshape(NB, H, B, D).reshape(NB, H, B, D // 4, 4).transpose(0, 1, 3, 2, 4)
    return f(inp["qkv"])[:, 0], np.ascontiguousarray(kc), f(inp["value_cache"])


@pytest.mark.parametrize("entry", ["rows", "partitioned"])
@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
def test_decode_is_no_further_from_exact_than_the_reference_kernels_own_arithmetic(pkg, dtype, entry):
    """The oracle's mode 1 follows the reference CUDA kernel's rounding points (weights rounded to the
    storage type, products and 8-wide partial sums in it, csrc/attention/attention_kernels.cu:236-254);
    mode 0 on float32 copies of the inputs is the exact result.  This kernel rounds less (fp16: nothing
    before the output; bfloat16: the weights once per block, products exact), so on unit-scale inputs its
    error against the exact result must not exceed the reference arithmetic's own in the root mean
    square, and its worst element (where the roundings that remain happen to line up: a noisy statistic
    over 12 288 elements) stays within 1.5 times the reference arithmetic's worst.  A second net under the
    per-element bfloat16 bound of decode_atol."""
    case = datagen.DecodeCase(name="closer", dtype=dtype, num_seqs=24, num_heads=4, head_size=128, block_size=16,
                              num_blocks=24 * 45 + 3, max_context=700, dist="normal", tables="permutation", seed=77)
    inp = datagen.make_decode_inputs(case)
    q32, kc32, vc32 = _storage_inputs_as_float32(inp, dtype)
    exact = pa_oracle.single_query_cached_kv_attention(q32, kc32, vc32, inp["scale"], inp["block_tables"],
                                                       inp["context_lens"], "float32").astype(np.float64)
    ref_kernel = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], dtype, mode=1), dtype).astype(np.float64)
    got = helpers.out_to_f32(_run_capi(pkg, inp, case, split=entry == "partitioned"), dtype).astype(np.float64)
    err_ref, err_got = np.abs(ref_kernel - exact), np.abs(got - exact)
    rms = lambda e: float(np.sqrt(np.mean(e * e)))  # noqa: E731
    assert rms(err_got) <= 1.02 * rms(err_ref), (rms(err_got), rms(err_ref))
    assert err_got.max() <= 1.5 * err_ref.max(), (err_got.max(), err_ref.max())


# ---- per-GPU shapes of the tensor-parallel configurations ---------------------

def _run_entry(pkg, entry, inp, case, ws_fill=float("nan")):
    """One decode call through the C ABI by entry-point name: "rows" (no
    workspace), "ws" (per-call plan), "planned" (plan made, then reused by a
    second call whose result is the one returned)."""
    import torch
    dev = "cuda:0"
    qkv, query, kc, vc, bt, cl = helpers.decode_inputs_to_device(inp, case.dtype, dev)
    out = torch.full((case.num_seqs, case.num_heads, case.head_size), float("nan"), dtype=qkv.dtype, device=dev)
    B, mc = case.block_size, inp["max_context_len"]
    if entry == "rows":
        pkg._capi.single_query_cached_kv_attention(out, query, kc, vc, inp["scale"], bt, cl, B, mc)
    else:
        nbytes = pkg._capi.decode_workspace_bytes(case.num_seqs, case.num_heads, case.head_size, B, mc)
        ws = torch.full((max(nbytes, 4) // 4,), ws_fill, dtype=torch.float32, device=dev) if nbytes else None
        if entry == "ws":
            pkg._capi.single_query_cached_kv_attention_ws(out, query, kc, vc, inp["scale"], bt, cl, B, mc, ws)
        else:
            total = int(np.asarray(inp["context_lens"], dtype=np.int64).sum())
            pkg._capi.single_query_cached_kv_attention_planned(out, query, kc, vc, inp["scale"], bt, cl, B, mc, ws,
                                                               False, total)
            out.fill_(float("nan"))
            pkg._capi.single_query_cached_kv_attention_planned(out, query, kc, vc, inp["scale"], bt, cl, B, mc, ws,
                                                               True, total)
    torch.cuda.synchronize()
    return out


_INPUT_CACHE = {}


def _inputs_once(case):
    """Inputs (and the oracle's result) of a case, generated once per process:
    the entry-point variants of one shape share them."""
    key = case.to_json()
    if key not in _INPUT_CACHE:
        _INPUT_CACHE.clear()  # one shape at a time: the caches are hundreds of megabytes
        inp = datagen.make_decode_inputs(case)
        _INPUT_CACHE[key] = (inp, _oracle_f32(inp, case), helpers.decode_atol(inp, case.dtype))
    return _INPUT_CACHE[key]


def _oracle_f32(inp, case):
    return datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], case.dtype), case.dtype)


@pytest.mark.parametrize("entry", ["rows", "ws", "planned"])
@pytest.mark.parametrize("dtype", datagen.DTYPES)
def test_decode_llama13b_tp2_shard_shape(pkg, dtype, entry):
    """LLaMA-13B over two GPUs: each rank attends 20 of the 40 heads, head size
    128 (vllm/model_executor/models/llama.py:86-89).  64 sequences, ragged
    contexts up to 1024 tokens, distinct physical blocks."""
    case = datagen.DecodeCase(name="c3", dtype=dtype, num_seqs=64, num_heads=20, head_size=128, block_size=16,
                              num_blocks=2400, max_context=1024, dist="normal", tables="permutation", seed=41)
    inp, ref, atol = _inputs_once(case)
    assert int(inp["context_lens"].max()) > 960 and int(inp["context_lens"].min()) < 64
    got = helpers.out_to_f32(_run_entry(pkg, entry, inp, case), dtype)
    helpers.assert_close(got, ref, atol, 2 * helpers.TIGHT[dtype][1], f"13B/TP2 shard {dtype} {entry}")


@pytest.mark.parametrize("entry", ["rows", "ws", "planned"])
@pytest.mark.parametrize("dtype", datagen.DTYPES)
def test_decode_neox20b_tp4_shard_shape_with_forked_tables(pkg, dtype, entry):
    """GPT-NeoX-20B over four GPUs with three samples per prompt: 16 of 64 heads
    per rank, head size 96 (gpt_neox.py:51-53); 96 sequences = 32 groups of 3
    that map the same physical blocks for their prompt's full blocks and
    distinct ones after (vllm/core/block_manager.py:111-135).  Exercises the
    per-block load policy for blocks a neighbouring sequence also maps."""
    case = datagen.DecodeCase(name="c4", dtype=dtype, num_seqs=96, num_heads=16, head_size=96, block_size=16,
                              num_blocks=2200, max_context=600, dist="normal", tables="forked", fork=3, seed=43)
    inp, ref, atol = _inputs_once(case)
    bt = inp["block_tables"]
    shared = sum(int(bt[s, 0] == bt[s + 1, 0]) for s in range(0, 96, 3) if inp["context_lens"][s] > 48)
    assert shared >= 20, "the forked tables share no blocks: the test would not exercise the shared-block path"
    got = helpers.out_to_f32(_run_entry(pkg, entry, inp, case), dtype)
    helpers.assert_close(got, ref, atol, 2 * helpers.TIGHT[dtype][1], f"NeoX-20B/TP4 shard n=3 {dtype} {entry}")


@pytest.mark.parametrize("entry", ["rows", "planned"])
def test_decode_forked_long_contexts_item_list(pkg, entry):
    """Forked groups whose shared prefix crosses the 1024-token partition edge:
    the item-list form reads shared blocks from listed partitions too."""
    ctx = []
    for g in range(24):
        base = [300, 1500, 2300, 40][g % 4] + 7 * g
        ctx += [base, base + 3, base + 1]
    case = datagen.DecodeCase(name="c4l", dtype="float16", num_seqs=len(ctx), num_heads=8, head_size=96, block_size=16,
                              num_blocks=sum((c + 15) // 16 for c in ctx) + 4, context_lens=ctx, dist="normal",
                              tables="forked", fork=3, seed=47)
    inp = datagen.make_decode_inputs(case)
    got = helpers.out_to_f32(_run_entry(pkg, entry, inp, case), case.dtype)
    atol, rtol = helpers.TIGHT[case.dtype]
    helpers.assert_close(got, _oracle_f32(inp, case), atol, 2 * rtol, f"forked long {entry}")


def test_decode_opt125m_microbench_size_properties(pkg):
    """The microbenchmark shape at full size: 256 sequences x 12 heads x 64,
    block 16, context 1024, fp16, distinct physical blocks.  Softmax weights sum
    to one (V == 1), and sampled sequences equal the oracle with all 12 heads,
    through the row form and the planned form."""
    import torch
    dev = "cuda:0"
    S, H, D, B, CTX = 256, 12, 64, 16, 1024
    nb = S * CTX // B
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    qkv = torch.randn(S, 3, H, D, dtype=torch.float16, device=dev, generator=g) * 0.5
    query = qkv[:, 0]
    kc = torch.randn(nb, H, D // 8, B, 8, dtype=torch.float16, device=dev, generator=g) * 0.5
    vc = torch.ones(nb, H, D, B, dtype=torch.float16, device=dev)
    perm = torch.randperm(nb, device=dev, generator=g).to(torch.int32).reshape(S, CTX // B)
    ctx_list = [CTX] * S
    ctx_list[7], ctx_list[8], ctx_list[200] = 1, CTX - 1, 513
    cl = torch.tensor(ctx_list, dtype=torch.int32, device=dev)
    scale = 1.0 / D ** 0.5
    nbytes = pkg._capi.decode_workspace_bytes(S, H, D, B, CTX)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=dev)

    def run(form, out):
        if form == "rows":
            pkg._capi.single_query_cached_kv_attention(out, query, kc, vc, scale, perm, cl, B, CTX)
        else:
            pkg._capi.single_query_cached_kv_attention_planned(out, query, kc, vc, scale, perm, cl, B, CTX,
                                                               ws if nbytes else None, False, sum(ctx_list))
        torch.cuda.synchronize()

    out = torch.empty(S, H, D, dtype=torch.float16, device=dev)
    for form in ("rows", "planned"):
        out.fill_(float("nan"))
        run(form, out)
        assert torch.allclose(out.float(), torch.ones_like(out, dtype=torch.float32), atol=1e-3), form
    vc.copy_(torch.randn(nb, H, D, B, dtype=torch.float16, device=dev, generator=g) * 0.5)
    for form in ("rows", "planned"):
        out.fill_(float("nan"))
        run(form, out)
        for s in (0, 7, 8, 200, 255):
            blocks = perm[s].long()
            ref = pa_oracle.single_query_cached_kv_attention(
                query[s:s + 1].contiguous().cpu().numpy(), kc[blocks].contiguous().cpu().numpy(),
                vc[blocks].contiguous().cpu().numpy(), scale, np.arange(len(blocks), dtype=np.int32)[None, :],
                np.array([ctx_list[s]], dtype=np.int32), "float16")
            helpers.assert_close(out[s].float().cpu().numpy(), ref.astype(np.float32)[0], 2e-5, 1.5e-3,
                                 f"microbench shape, {form}, sequence {s}")
