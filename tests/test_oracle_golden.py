"""CPU tests: the C oracle against the golden fixtures that were produced by
the reference's own Python statements of the path (oracle/gen_golden.py)."""
import hashlib

import numpy as np
import pytest

from oracle import datagen, pa_oracle
from tests import helpers

DECODE_META, DECODE_DATA = helpers.load_decode_golden()
CACHE_META, CACHE_DATA = helpers.load_cache_golden()


def _digest(arrs) -> str:
    h = hashlib.sha256()
    for a in arrs:
        a = np.ascontiguousarray(a)
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


# ---- scalar conversions -------------------------------------------------------

def test_half_conversion_matches_numpy_for_all_bit_patterns():
    L = pa_oracle.lib()
    bits = np.arange(65536, dtype=np.uint16)
    want = bits.view(np.float16).astype(np.float32)
    for b in range(0, 65536, 7):  # every 7th pattern incl. subnormals, inf, nan
        got = L.pao_test_half_to_float(int(b))
        w = float(want[b])
        assert (np.isnan(got) and np.isnan(w)) or got == w, (b, got, w)


def test_float_to_half_rounding_matches_numpy():
    L = pa_oracle.lib()
    vals = np.concatenate([
        datagen.normal(4000, 7, 1.0), datagen.normal(2000, 8, 1e-4), datagen.normal(2000, 9, 3e4),
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, 5.96e-8, 2.98e-8, 2.99e-8, 6.1e-5, 6.0e-5])
    ]).astype(np.float32)
    want = vals.astype(np.float16).view(np.uint16)
    for v, w in zip(vals, want):
        assert L.pao_test_float_to_half(float(v)) == int(w), (float(v), int(w))


def test_bf16_rounding_matches_torch():
    import torch
    L = pa_oracle.lib()
    vals = np.concatenate([datagen.normal(3000, 17, 1.0), datagen.normal(1000, 18, 1e-20),
                           datagen.normal(1000, 19, 1e20)]).astype(np.float32)
    want = torch.from_numpy(vals).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    got_np = datagen.float_to_bf16_bits(vals)
    assert np.array_equal(got_np, want)
    for v, w in zip(vals[:500], want[:500]):
        assert L.pao_test_float_to_bf16(float(v)) == int(w)


# ---- decode attention ------------------------------------------------------------

def _small(name: str) -> bool:
    return not name.startswith("reftest_")


@pytest.mark.parametrize("name", sorted(DECODE_META))
def test_decode_oracle_matches_reference_fixture(name):
    entry = DECODE_META[name]
    case = helpers.decode_case_from_meta(entry)
    inp = datagen.make_decode_inputs(case)
    assert datagen.checksum(inp) == entry["inputs_sha256"], "regenerated inputs differ from the fixture's"
    out = pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], case.dtype, mode=0)
    got = datagen.storage_to_float32(out, case.dtype)
    # (a) the reference formula evaluated by the reference's function in fp32
    atol, rtol = helpers.TIGHT[case.dtype]
    helpers.assert_close(got, DECODE_DATA[name + "/ref_f32"], atol, rtol, f"{name} vs reference fp32")
    # (b) the reference function run in the storage type, with the reference
    #     test's tolerance; only meaningful for its own input distribution
    #     (at unit scale the fp16/bf16 evaluation of the reference itself is
    #     off by more than 1e-3)
    if case.dist == "uniform1e-3":
        want = datagen.storage_to_float32(DECODE_DATA[name + "/ref_dtype"], case.dtype)
        helpers.assert_close(got, want, helpers.REF_ATOL, helpers.REF_RTOL, f"{name} vs reference in {case.dtype}")


@pytest.mark.parametrize("name", sorted(n for n in DECODE_META if _small(n)))
def test_decode_oracle_kernel_arithmetic_mode_stays_within_reference_tolerance(name):
    """Mode 1 rounds where the CUDA kernel rounds (P in the storage type,
    8-wide partial sums in it).  It must agree with the exact evaluation
    within the parity bar of 1e-3 that the GPU kernel is held to, scaled by
    the output magnitude for unit-scale inputs."""
    entry = DECODE_META[name]
    case = helpers.decode_case_from_meta(entry)
    inp = datagen.make_decode_inputs(case)
    args = (inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
            inp["context_lens"], case.dtype)
    exact = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(*args, mode=0), case.dtype)
    kern = datagen.storage_to_float32(pa_oracle.single_query_cached_kv_attention(*args, mode=1), case.dtype)
    rtol = {"float32": 1e-5, "float16": 4e-3, "bfloat16": 3e-2}[case.dtype]
    helpers.assert_close(kern, exact, 1e-3 if case.dist == "uniform1e-3" else 2e-3, rtol, name)


def test_decode_oracle_ignores_table_entries_past_the_context():
    case = datagen.DecodeCase(name="pad", dtype="float16", num_seqs=3, num_heads=2, head_size=64,
                              block_size=8, num_blocks=16, context_lens=[5, 9, 16], dist="normal",
                              table_pad=2, seed=5)
    inp = datagen.make_decode_inputs(case)
    a = pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], inp["block_tables"],
        inp["context_lens"], case.dtype)
    tables = inp["block_tables"].copy()
    for s, c in enumerate(inp["context_lens"]):
        tables[s, (c + 7) // 8:] = 15  # garbage in the padding
    b = pa_oracle.single_query_cached_kv_attention(
        inp["qkv"][:, 0], inp["key_cache"], inp["value_cache"], inp["scale"], tables,
        inp["context_lens"], case.dtype)
    assert np.array_equal(a, b)


# ---- cache ops ----------------------------------------------------------------------

@pytest.mark.parametrize("name", sorted(n for n in CACHE_META if n.startswith("reshape")))
def test_reshape_and_gather_oracle_match_reference_fixture(name):
    entry = CACHE_META[name]
    case = helpers.cache_case_from_meta(entry)
    inp = datagen.make_cache_inputs(case)
    kc, vc = inp["key_caches"][0].copy(), inp["value_caches"][0].copy()
    pa_oracle.reshape_and_cache(inp["qkv"][:, 1], inp["qkv"][:, 2], kc, vc, inp["slot_mapping"], case.dtype)
    assert _digest([kc]) == entry["reshape_key_cache_sha256"]
    assert _digest([vc]) == entry["reshape_value_cache_sha256"]
    qkv2 = inp["qkv"].copy()
    pa_oracle.gather_cached_kv(qkv2[:, 1], qkv2[:, 2], inp["key_caches"][0], inp["value_caches"][0],
                               inp["slot_mapping"], case.dtype)
    assert _digest([qkv2]) == entry["gather_qkv_sha256"]
    assert np.array_equal(qkv2[:, 0], inp["qkv"][:, 0]), "gather must not touch the query slice"
    if name + "/reshape_key_cache" in CACHE_DATA:
        assert np.array_equal(kc, CACHE_DATA[name + "/reshape_key_cache"])
        assert np.array_equal(vc, CACHE_DATA[name + "/reshape_value_cache"])
        assert np.array_equal(qkv2, CACHE_DATA[name + "/gather_qkv"])


@pytest.mark.parametrize("name", sorted(n for n in CACHE_META if n.startswith("copy")))
def test_copy_blocks_oracle_matches_reference_fixture(name):
    entry = CACHE_META[name]
    case = helpers.cache_case_from_meta(entry)
    inp = datagen.make_cache_inputs(case)
    kcs = [a.copy() for a in inp["key_caches"]]
    vcs = [a.copy() for a in inp["value_caches"]]
    pa_oracle.copy_blocks(kcs, vcs, inp["src_blocks"], inp["dst_blocks"])
    assert _digest(kcs) == entry["key_caches_sha256"]
    assert _digest(vcs) == entry["value_caches_sha256"]


def test_swap_blocks_oracle_is_a_block_copy():
    case = datagen.CacheCase(name="swap", dtype="float16", num_tokens=0, num_heads=3, head_size=32,
                             block_size=8, num_blocks=10, num_layers=2, num_mappings=4, seed=3)
    inp = datagen.make_cache_inputs(case)
    src, dst = inp["key_caches"][0], inp["key_caches"][1].copy()
    before = dst.copy()
    pa_oracle.swap_blocks(src, dst, inp["src_blocks"], inp["dst_blocks"])
    touched = set(int(d) for d in inp["dst_blocks"])
    for s, d in zip(inp["src_blocks"], inp["dst_blocks"]):
        assert np.array_equal(dst[d], src[s])
    for b in range(case.num_blocks):
        if b not in touched:
            assert np.array_equal(dst[b], before[b])
