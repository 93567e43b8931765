"""GPU parity tests of the cache kernels (bit-exact), driven through the C ABI
and through the `cache_ops` module."""
import hashlib

import numpy as np
import pytest

from oracle import datagen, pa_oracle
from tests import helpers

pytestmark = pytest.mark.gpu

CACHE_META, CACHE_DATA = helpers.load_cache_golden()
DEV = "cuda:0"


def _digest(arrs) -> str:
    h = hashlib.sha256()
    for a in arrs:
        a = np.ascontiguousarray(a)
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def _to_dev(a, dtype):
    return datagen.storage_to_torch(np.ascontiguousarray(a).copy(), dtype).to(DEV)


@pytest.mark.parametrize("name", sorted(n for n in CACHE_META if n.startswith("reshape")))
def test_reshape_and_gather_match_reference_fixture(pkg, name):
    import torch
    entry = CACHE_META[name]
    case = helpers.cache_case_from_meta(entry)
    inp = datagen.make_cache_inputs(case)
    qkv = _to_dev(inp["qkv"], case.dtype)
    _, key, value = qkv.unbind(dim=1)
    kc, vc = _to_dev(inp["key_caches"][0], case.dtype), _to_dev(inp["value_caches"][0], case.dtype)
    slots = torch.from_numpy(inp["slot_mapping"]).to(DEV)
    pkg._capi.reshape_and_cache(key, value, kc, vc, slots)
    torch.cuda.synchronize()
    assert _digest([datagen.torch_to_storage(kc, case.dtype)]) == entry["reshape_key_cache_sha256"]
    assert _digest([datagen.torch_to_storage(vc, case.dtype)]) == entry["reshape_value_cache_sha256"]
    # gather from the ORIGINAL caches into a fresh qkv
    qkv2 = _to_dev(inp["qkv"], case.dtype)
    _, key2, value2 = qkv2.unbind(dim=1)
    kc0, vc0 = _to_dev(inp["key_caches"][0], case.dtype), _to_dev(inp["value_caches"][0], case.dtype)
    pkg._capi.gather_cached_kv(key2, value2, kc0, vc0, slots)
    torch.cuda.synchronize()
    assert _digest([datagen.torch_to_storage(qkv2, case.dtype)]) == entry["gather_qkv_sha256"]


@pytest.mark.parametrize("name", sorted(n for n in CACHE_META if n.startswith("copy")))
def test_copy_blocks_matches_reference_fixture(pkg, name):
    import torch
    entry = CACHE_META[name]
    case = helpers.cache_case_from_meta(entry)
    inp = datagen.make_cache_inputs(case)
    kcs = [_to_dev(a, case.dtype) for a in inp["key_caches"]]
    vcs = [_to_dev(a, case.dtype) for a in inp["value_caches"]]
    mapping = {int(s): [int(d)] for s, d in zip(inp["src_blocks"], inp["dst_blocks"])}
    pkg.cache_ops.copy_blocks(kcs, vcs, mapping)
    torch.cuda.synchronize()
    assert _digest(datagen.torch_to_storage(t, case.dtype) for t in kcs) == entry["key_caches_sha256"]
    assert _digest(datagen.torch_to_storage(t, case.dtype) for t in vcs) == entry["value_caches_sha256"]


@pytest.mark.parametrize("dtype", datagen.DTYPES)
def test_cache_ops_module_like_the_reference_tests(pkg, dtype):
    """tests/kernels/test_cache.py:66-146 (reshape_and_cache, gather_cached_kv)
    at the reference's sizes, through the pybind11 module, checked against the
    oracle bit for bit."""
    import torch
    case = datagen.CacheCase(name="ref", dtype=dtype, num_tokens=3, num_heads=2, head_size=16, block_size=8,
                             num_blocks=2, seed=77)
    inp = datagen.make_cache_inputs(case)
    qkv = _to_dev(inp["qkv"], dtype)
    _, key, value = qkv.unbind(dim=1)
    kc, vc = _to_dev(inp["key_caches"][0], dtype), _to_dev(inp["value_caches"][0], dtype)
    slots = torch.from_numpy(inp["slot_mapping"]).to(DEV)
    pkg.cache_ops.reshape_and_cache(key, value, kc, vc, slots)
    kc_ref, vc_ref = inp["key_caches"][0].copy(), inp["value_caches"][0].copy()
    pa_oracle.reshape_and_cache(inp["qkv"][:, 1], inp["qkv"][:, 2], kc_ref, vc_ref, inp["slot_mapping"], dtype)
    torch.cuda.synchronize()
    assert np.array_equal(datagen.torch_to_storage(kc, dtype), kc_ref)
    assert np.array_equal(datagen.torch_to_storage(vc, dtype), vc_ref)
    # gather back: key/value rows must come back unchanged, query untouched
    qkv2 = torch.zeros_like(qkv)
    qkv2[:, 0] = qkv[:, 0]
    _, k2, v2 = qkv2.unbind(dim=1)
    pkg.cache_ops.gather_cached_kv(k2, v2, kc, vc, slots)
    torch.cuda.synchronize()
    assert torch.equal(qkv2, qkv)


@pytest.mark.parametrize("dtype", datagen.DTYPES)
def test_reshape_and_cache_odd_shapes_and_unaligned_rows(pkg, dtype):
    import torch
    case = datagen.CacheCase(name="odd", dtype=dtype, num_tokens=19, num_heads=5, head_size=80, block_size=32,
                             num_blocks=7, seed=5)
    inp = datagen.make_cache_inputs(case)
    T, H, D = case.num_tokens, case.num_heads, case.head_size
    tdt = datagen.storage_to_torch(inp["qkv"], dtype).dtype
    # token rows at an odd element offset and an odd row stride -> scalar path
    stride = H * D + 1
    kbuf = torch.zeros(T * stride + 1, dtype=tdt, device=DEV)
    vbuf = torch.zeros(T * stride + 1, dtype=tdt, device=DEV)
    key = kbuf[1:].as_strided((T, H, D), (stride, D, 1))
    value = vbuf[1:].as_strided((T, H, D), (stride, D, 1))
    qkv = _to_dev(inp["qkv"], dtype)
    key.copy_(qkv[:, 1])
    value.copy_(qkv[:, 2])
    kc, vc = _to_dev(inp["key_caches"][0], dtype), _to_dev(inp["value_caches"][0], dtype)
    slots = torch.from_numpy(inp["slot_mapping"]).to(DEV)
    pkg._capi.reshape_and_cache(key, value, kc, vc, slots)
    torch.cuda.synchronize()
    kc_ref, vc_ref = inp["key_caches"][0].copy(), inp["value_caches"][0].copy()
    pa_oracle.reshape_and_cache(inp["qkv"][:, 1], inp["qkv"][:, 2], kc_ref, vc_ref, inp["slot_mapping"], dtype)
    assert np.array_equal(datagen.torch_to_storage(kc, dtype), kc_ref)
    assert np.array_equal(datagen.torch_to_storage(vc, dtype), vc_ref)
    # and back through the scalar gather path
    kbuf.zero_()
    vbuf.zero_()
    pkg._capi.gather_cached_kv(key, value, kc, vc, slots)
    torch.cuda.synchronize()
    assert torch.equal(key, qkv[:, 1]) and torch.equal(value, qkv[:, 2])


def test_reshape_and_cache_skips_negative_slots_and_empty_input(pkg):
    import torch
    case = datagen.CacheCase(name="neg", dtype="float16", num_tokens=4, num_heads=2, head_size=64, block_size=16,
                             num_blocks=3, seed=9)
    inp = datagen.make_cache_inputs(case)
    qkv = _to_dev(inp["qkv"], "float16")
    kc, vc = _to_dev(inp["key_caches"][0], "float16"), _to_dev(inp["value_caches"][0], "float16")
    slots_np = inp["slot_mapping"].copy()
    slots_np[2] = -1
    slots = torch.from_numpy(slots_np).to(DEV)
    pkg.cache_ops.reshape_and_cache(qkv[:, 1], qkv[:, 2], kc, vc, slots)
    torch.cuda.synchronize()
    kc_ref, vc_ref = inp["key_caches"][0].copy(), inp["value_caches"][0].copy()
    keep = [0, 1, 3]
    pa_oracle.reshape_and_cache(np.ascontiguousarray(inp["qkv"][keep, 1]), np.ascontiguousarray(inp["qkv"][keep, 2]),
                                kc_ref, vc_ref, slots_np[keep], "float16")
    assert np.array_equal(datagen.torch_to_storage(kc, "float16"), kc_ref)
    assert np.array_equal(datagen.torch_to_storage(vc, "float16"), vc_ref)
    empty = qkv[:0]
    pkg.cache_ops.reshape_and_cache(empty[:, 1], empty[:, 2], kc, vc, slots[:0])
    torch.cuda.synchronize()


def test_copy_blocks_many_pairs_and_layers_in_several_launches(pkg):
    """More layers and pairs than one kernel-argument table holds, one source
    fanning out to several destinations (copy-on-write for n > 2 samples)."""
    import torch
    L, NB, H, D, B = 100, 700, 2, 16, 8
    g = torch.Generator(device=DEV)
    g.manual_seed(1)
    kcs = [torch.randn(NB, H, D // 8, B, 8, dtype=torch.float16, device=DEV, generator=g) for _ in range(L)]
    vcs = [torch.randn(NB, H, D, B, dtype=torch.float16, device=DEV, generator=g) for _ in range(L)]
    perm = datagen.permutation(NB, 5)
    srcs, dsts = perm[:100], perm[100:400]
    mapping = {int(s): [int(d) for d in dsts[3 * i:3 * i + 3]] for i, s in enumerate(srcs)}
    ref_k = [t.cpu().numpy().copy() for t in kcs]
    ref_v = [t.cpu().numpy().copy() for t in vcs]
    flat_src = [s for s, ds in mapping.items() for _ in ds]
    flat_dst = [d for _, ds in mapping.items() for d in ds]
    pa_oracle.copy_blocks(ref_k, ref_v, flat_src, flat_dst)
    pkg.cache_ops.copy_blocks(kcs, vcs, mapping)
    torch.cuda.synchronize()
    for l in range(L):
        assert np.array_equal(kcs[l].cpu().numpy(), ref_k[l]), l
        assert np.array_equal(vcs[l].cpu().numpy(), ref_v[l]), l


def test_copy_blocks_narrow_word_paths(pkg):
    """Block sizes that are not multiples of 16 bytes fall back to narrower
    words (driven through the C ABI with byte tensors)."""
    import torch
    for block_bytes in (24, 36, 6, 7):
        NB = 9
        k = torch.arange(NB * block_bytes, dtype=torch.int64, device=DEV).to(torch.uint8).reshape(NB, block_bytes)
        v = (k + 101).to(torch.uint8)
        k0, v0 = k.clone(), v.clone()
        pkg._capi.copy_blocks([k], [v], [0, 3], [5, 8])
        torch.cuda.synchronize()
        for s, d in ((0, 5), (3, 8)):
            assert torch.equal(k[d], k0[s]) and torch.equal(v[d], v0[s])
        untouched = [0, 1, 2, 3, 4, 6, 7]
        assert torch.equal(k[untouched], k0[untouched]) and torch.equal(v[untouched], v0[untouched])


def test_swap_blocks_gpu_cpu_round_trip_on_side_stream(pkg):
    """CacheEngine._swap (vllm/worker/cache_engine.py:101-118): swap out to
    pinned CPU memory and back in on a side stream, fenced by an event."""
    import torch
    NB_GPU, NB_CPU, H, D, B = 12, 6, 3, 64, 16
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    gpu = torch.randn(NB_GPU, H, D // 8, B, 8, dtype=torch.float16, device=DEV, generator=g)
    gpu0 = gpu.clone()
    cpu = torch.zeros(NB_CPU, H, D // 8, B, 8, dtype=torch.float16).pin_memory()
    out_map = {1: 0, 2: 1, 3: 2, 7: 5, 11: 4}   # contains a mergeable run 1,2,3 -> 0,1,2
    stream = torch.cuda.Stream()
    event = torch.cuda.Event()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        pkg.cache_ops.swap_blocks(gpu, cpu, out_map)
        event.record(stream=stream)
    event.synchronize()
    for s, d in out_map.items():
        assert torch.equal(cpu[d], gpu0[s].cpu())
    assert torch.count_nonzero(cpu[3]) == 0
    gpu.zero_()
    in_map = {d: s for s, d in out_map.items()}
    with torch.cuda.stream(stream):
        pkg.cache_ops.swap_blocks(cpu, gpu, in_map)
        event.record(stream=stream)
    event.wait()
    torch.cuda.synchronize()
    for s in out_map:
        assert torch.equal(gpu[s], gpu0[s])
    # device-to-device between two cache tensors
    other = torch.zeros_like(gpu)
    pkg.cache_ops.swap_blocks(gpu0, other, {0: 11, 5: 6})
    torch.cuda.synchronize()
    assert torch.equal(other[11], gpu0[0]) and torch.equal(other[6], gpu0[5])
    with pytest.raises(RuntimeError, match="Invalid device combination"):
        pkg.cache_ops.swap_blocks(cpu, cpu.clone(), {0: 1})


def test_reshape_gather_round_trip_at_decode_step_size(pkg):
    """LLaMA-7B decode step: 256 tokens x 32 heads x 128, block 16.  Scatter
    then gather must return the rows bit for bit, and only the addressed slots
    of the caches may change."""
    import torch
    T, H, D, B, NB = 256, 32, 128, 16, 4096
    g = torch.Generator(device=DEV)
    g.manual_seed(4)
    qkv = torch.randn(T, 3, H, D, dtype=torch.float16, device=DEV, generator=g)
    kc = torch.zeros(NB, H, D // 8, B, 8, dtype=torch.float16, device=DEV)
    vc = torch.zeros(NB, H, D, B, dtype=torch.float16, device=DEV)
    slots = torch.randperm(NB * B, device=DEV, generator=g)[:T].to(torch.int32)
    pkg.cache_ops.reshape_and_cache(qkv[:, 1], qkv[:, 2], kc, vc, slots)
    out = torch.zeros_like(qkv)
    pkg.cache_ops.gather_cached_kv(out[:, 1], out[:, 2], kc, vc, slots)
    torch.cuda.synchronize()
    assert torch.equal(out[:, 1], qkv[:, 1]) and torch.equal(out[:, 2], qkv[:, 2])
    assert torch.count_nonzero(out[:, 0]) == 0
    # checksum of checksums: total mass moved equals the mass of the inputs
    assert torch.equal(kc.double().sum(), qkv[:, 1].double().sum())
    assert torch.equal(vc.double().sum(), qkv[:, 2].double().sum())
    assert int(torch.count_nonzero(kc)) <= T * H * D and int(torch.count_nonzero(vc)) <= T * H * D


def _prompt_like_slots(rng, lens, B, NB, T_pad):
    """Consecutive slots per prompt through a scattered block table, padding rows = -1."""
    blocks = rng.permutation(NB)
    slots, nxt = [], 0
    for n in lens:
        nb = (n + B - 1) // B
        table = blocks[nxt:nxt + nb]
        nxt += nb
        slots += [int(table[i // B]) * B + i % B for i in range(n)]
    slots += [-1] * (T_pad - len(slots))
    return np.asarray(slots, dtype=np.int32)


@pytest.mark.parametrize("dtype,H,D,B", [("float16", 4, 128, 16), ("bfloat16", 3, 80, 32), ("float32", 2, 64, 8),
                                         ("float16", 5, 96, 8), ("float32", 3, 128, 16)])
@pytest.mark.parametrize("pattern", ["prompts", "singles", "mixed"])
def test_reshape_and_cache_prompt_sized_batches_match_oracle(pkg, dtype, H, D, B, pattern):
    """Batches large enough for the run-aware kernel: prompt lengths that leave
    runs unaligned to both the 16-token chunks and the blocks, isolated tokens,
    and a mix with skipped (negative) slots.  Bit-exact against the oracle on the
    whole caches (untouched slots keep their previous contents)."""
    import torch
    rng = np.random.default_rng(hash((dtype, H, D, B, pattern)) % (2 ** 32))
    NB = 160
    if pattern == "prompts":
        lens = [5, 17, 33, 64, 1, 130, 255, 16, 47, 71]
        T = sum(lens) + 9
        slots = _prompt_like_slots(rng, lens, B, NB, T)
    elif pattern == "singles":
        T = 600
        slots = rng.permutation(NB * B)[:T].astype(np.int32)
    else:
        lens = [40, 3, 200, 90]
        T = 700
        slots = _prompt_like_slots(rng, lens, B, NB // 2, sum(lens))
        extra = (NB // 2) * B + rng.permutation((NB // 2) * B)[:T - len(slots)].astype(np.int32)
        extra[::7] = -1
        slots = np.concatenate([slots, extra]).astype(np.int32)
    assert T >= 512
    case = datagen.CacheCase(name="runs", dtype=dtype, num_tokens=T, num_heads=H, head_size=D, block_size=B,
                             num_blocks=NB, seed=77)
    inp = datagen.make_cache_inputs(case)
    qkv = inp["qkv"]
    kc_ref, vc_ref = inp["key_caches"][0].copy(), inp["value_caches"][0].copy()
    valid = np.flatnonzero(slots >= 0)  # the reference has no notion of skipped tokens: leave them out
    pa_oracle.reshape_and_cache(np.ascontiguousarray(qkv[valid, 1]), np.ascontiguousarray(qkv[valid, 2]), kc_ref, vc_ref,
                                slots[valid], dtype)
    qkv_d = _to_dev(qkv, dtype)
    kc, vc = _to_dev(inp["key_caches"][0], dtype), _to_dev(inp["value_caches"][0], dtype)
    pkg.cache_ops.reshape_and_cache(qkv_d[:, 1], qkv_d[:, 2], kc, vc, torch.from_numpy(slots).to(DEV))
    torch.cuda.synchronize()
    assert np.array_equal(datagen.torch_to_storage(kc, dtype), kc_ref)
    assert np.array_equal(datagen.torch_to_storage(vc, dtype), vc_ref)
    # and back: gather returns the rows of every written token
    out = torch.zeros_like(qkv_d)
    pkg.cache_ops.gather_cached_kv(out[:, 1], out[:, 2], kc, vc, torch.from_numpy(slots).to(DEV))
    torch.cuda.synchronize()
    keep = torch.from_numpy(slots >= 0).to(DEV)
    assert torch.equal(out[keep][:, 1:], qkv_d[keep][:, 1:])
    assert torch.count_nonzero(out[~keep]) == 0 and torch.count_nonzero(out[:, 0]) == 0  # nothing else is touched


def test_reshape_and_cache_at_prompt_batch_size_round_trip(pkg):
    """A 16k-token LLaMA-7B prompt batch (32 heads x 128, block 16): scatter,
    gather back, and compare the value cache with torch's own indexing."""
    import torch
    T, H, D, B = 16384, 32, 128, 16
    NB = T // B + 64
    g = torch.Generator(device=DEV)
    g.manual_seed(9)
    qkv = torch.randn(T, 3, H, D, dtype=torch.float16, device=DEV, generator=g)
    kc = torch.zeros(NB, H, D // 8, B, 8, dtype=torch.float16, device=DEV)
    vc = torch.zeros(NB, H, D, B, dtype=torch.float16, device=DEV)
    table = torch.randperm(NB, device=DEV, generator=g)[:T // B]
    pos = torch.arange(T, device=DEV)
    slots = (table[pos // B] * B + pos % B).to(torch.int32)
    pkg.cache_ops.reshape_and_cache(qkv[:, 1], qkv[:, 2], kc, vc, slots)
    out = torch.zeros_like(qkv)
    pkg.cache_ops.gather_cached_kv(out[:, 1], out[:, 2], kc, vc, slots)
    torch.cuda.synchronize()
    assert torch.equal(out[:, 1:], qkv[:, 1:])
    want_v = torch.zeros_like(vc)
    want_v[(slots // B).long(), :, :, (slots % B).long()] = qkv[:, 2]
    assert torch.equal(vc, want_v)
    want_k = torch.zeros_like(kc)
    want_k[(slots // B).long(), :, :, (slots % B).long(), :] = qkv[:, 1].reshape(T, H, D // 8, 8)
    assert torch.equal(kc, want_k)


@pytest.mark.parametrize("pinned", [True, False], ids=["pinned", "pageable"])
def test_swap_blocks_many_scattered_blocks(pkg, pinned):
    """Enough scattered blocks for the one-launch kernel path (pinned host memory
    is addressed by the device directly); pageable host memory cannot be and takes
    the per-run copies.  Both directions, on a side stream, untouched blocks intact."""
    import torch
    H, D, B, NBG, NBC, n = 4, 64, 16, 300, 120, 70
    g = torch.Generator(device="cpu").manual_seed(5)
    gpu = torch.randn(NBG, H, D // 8, B, 8, dtype=torch.float16, device=DEV)
    cpu = torch.full((NBC, H, D // 8, B, 8), -1.0, dtype=torch.float16)
    if pinned:
        cpu = cpu.pin_memory()
    src = torch.randperm(NBG, generator=g)[:n].tolist()
    dst = torch.randperm(NBC, generator=g)[:n].tolist()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        pkg.cache_ops.swap_blocks(gpu, cpu, dict(zip(src, dst)))
    side.synchronize()
    want = torch.full_like(cpu, -1.0)
    want[dst] = gpu[src].cpu()
    assert torch.equal(cpu, want)
    # and back into other GPU blocks
    back = torch.randperm(NBG, generator=g)[:n].tolist()
    before = gpu.clone()
    with torch.cuda.stream(side):
        pkg.cache_ops.swap_blocks(cpu, gpu, dict(zip(dst, back)))
    side.synchronize()
    before[back] = cpu[dst].to(DEV)
    assert torch.equal(gpu, before)


@pytest.mark.parametrize("dtype,H,D,B", [("float16", 32, 128, 16), ("bfloat16", 5, 80, 8), ("float32", 3, 64, 32),
                                         ("float16", 12, 64, 32)])
def test_reshape_and_cache_decode_batches_rewrite_whole_value_tiles_safely(pkg, dtype, H, D, B):
    """Decode-sized batches (16..511 tokens) rewrite the whole value tile of a
    token's block — unless another token of the batch goes to the same block,
    in which case elements are stored.  A batch with both kinds, padding rows
    (slot -1) and a cache full of other data: the result must equal the oracle's
    bit for bit, i.e. nothing but the addressed slots may change."""
    import torch
    rng = np.random.default_rng(31)
    T, NB = 200, 300
    tdt = getattr(torch, dtype)
    es = torch.tensor([], dtype=tdt).element_size()
    x = 16 // es
    blocks = rng.permutation(NB)
    slots = []
    for t in range(T):
        if t < 120:
            slots.append(int(blocks[t]) * B + int(rng.integers(0, B)))        # alone in its block
        elif t < 180:
            slots.append(int(blocks[120 + (t - 120) // 3]) * B + (t - 120) % 3)  # three tokens per block
        else:
            slots.append(-1)
    order = rng.permutation(T)
    slots = [slots[i] for i in order]
    g = torch.Generator(device=DEV)
    g.manual_seed(32)
    qkv = torch.randn(T, 3, H, D, device=DEV, generator=g).to(tdt)
    kc = torch.randn(NB, H, D // x, B, x, device=DEV, generator=g).to(tdt)
    vc = torch.randn(NB, H, D, B, device=DEV, generator=g).to(tdt)
    view = {2: torch.int16, 4: torch.int32}[es]
    keep = [i for i, s in enumerate(slots) if s >= 0]
    kc_ref = kc.cpu().view(view).numpy().copy()
    vc_ref = vc.cpu().view(view).numpy().copy()
    pa_oracle.reshape_and_cache(qkv[keep, 1].cpu().contiguous().view(view).numpy(), qkv[keep, 2].cpu().contiguous().view(view).numpy(),
                                kc_ref, vc_ref, np.asarray([slots[i] for i in keep], dtype=np.int32),
                                {2: "float16", 4: "float32"}[es])
    pkg.cache_ops.reshape_and_cache(qkv[:, 1], qkv[:, 2], kc, vc, torch.tensor(slots, dtype=torch.int32, device=DEV))
    torch.cuda.synchronize()
    assert np.array_equal(kc.cpu().view(view).numpy(), kc_ref)
    assert np.array_equal(vc.cpu().view(view).numpy(), vc_ref)
