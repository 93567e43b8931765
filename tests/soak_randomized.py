#!/usr/bin/env python3
"""Longer differential run of the randomized parity tests: the same generators as
test_decode_randomized_shapes_against_oracle and test_kernel_randomized_shapes_against_oracle
(prompt attention), plus random shapes of the append form and of the cache operators, on seeds beyond the ones the test suite
holds.  Not collected by pytest.

    python tests/soak_randomized.py [first_seed] [count]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402


def append_case(pkg, seed):
    """Random shape: the one-launch cache write + attention against the two calls (bit for bit) and
    against the oracle, in a randomly chosen launch form."""
    import numpy as np
    from oracle import datagen
    from tests import test_append_decode_gpu as t
    rng = np.random.default_rng(5000 + seed)
    dtype = ["float16", "bfloat16", "float32"][seed % 3]
    D, B = int(rng.choice([64, 80, 96, 128])), int(rng.choice([8, 16, 32]))
    H, S = int(rng.choice([1, 2, 3, 5, 8])), int(rng.choice([1, 2, 7, 33, 70, 130]))
    ctx = [int(rng.integers(1, 40)) if rng.random() < 0.4 else int(rng.integers(40, 1400)) for _ in range(S)]
    if S * H * sum(ctx) * D > 2e8:
        ctx = [max(1, c // 4) for c in ctx]
    form = ["rows", "planned", "per_call"][int(rng.integers(0, 3))]
    case = datagen.DecodeCase(name=f"soak{seed}", dtype=dtype, num_seqs=S, num_heads=H, head_size=D, block_size=B,
                              num_blocks=sum((c + B - 1) // B for c in ctx) + 3, context_lens=ctx, dist="normal",
                              tables="permutation", table_pad=int(rng.integers(0, 3)), seed=6000 + seed)
    what = f"seed {seed} S={S} H={H} D={D} B={B} {dtype} {form}"
    inp, two, one = t._both_ways(pkg, case, form, total_hint=bool(rng.integers(0, 2)) and form == "planned")
    t._assert_identical(two, one, what)
    t._assert_oracle(inp, case, one, what)


def cache_case(pkg, seed):
    """Random token batch through reshape_and_cache (all three kernels: a few tokens, a decode-sized
    batch, a prompt-sized batch; scattered slots, runs of consecutive slots, tokens without a slot),
    gather_cached_kv and copy_blocks, bit for bit against the oracle."""
    import numpy as np
    import torch
    from oracle import datagen, pa_oracle
    rng = np.random.default_rng(9000 + seed)
    dtype = ["float16", "bfloat16", "float32"][seed % 3]
    D, B = int(rng.choice([64, 80, 96, 128])), int(rng.choice([8, 16, 32]))
    H = int(rng.choice([1, 2, 5, 8, 12]))
    T = int(rng.choice([1, 3, 15, 16, 40, 200, 511, 512, 700]))
    L = int(rng.choice([1, 2, 3]))
    NB = (T + B - 1) // B + int(rng.integers(2, 9))
    M = int(rng.integers(0, NB // 2 + 1))
    case = datagen.CacheCase(name=f"soak{seed}", dtype=dtype, num_tokens=T, num_heads=H, head_size=D, block_size=B,
                             num_blocks=NB, num_layers=L, num_mappings=M, seed=7000 + seed)
    inp = datagen.make_cache_inputs(case)
    slots = inp["slot_mapping"].copy()
    kind = int(rng.integers(0, 3))
    if kind == 1:    # prompt-like: runs of consecutive slots starting at random places
        order = rng.permutation(NB)
        slots = np.concatenate([np.arange(b * B, b * B + B) for b in order])[:T].astype(np.int32)
    elif kind == 2:  # some tokens without a slot
        slots[rng.random(T) < 0.2] = -1
    what = f"seed {seed} T={T} H={H} D={D} B={B} L={L} {dtype} slots kind {kind}"
    dev = "cuda:0"
    to_dev = lambda a: datagen.storage_to_torch(np.ascontiguousarray(a).copy(), dtype).to(dev)  # noqa: E731
    qkv = to_dev(inp["qkv"])
    kcs, vcs = [to_dev(a) for a in inp["key_caches"]], [to_dev(a) for a in inp["value_caches"]]
    slots_d = torch.from_numpy(slots).to(dev)
    pkg._capi.reshape_and_cache(qkv[:, 1], qkv[:, 2], kcs[0], vcs[0], slots_d)
    kref, vref = [a.copy() for a in inp["key_caches"]], [a.copy() for a in inp["value_caches"]]
    keep = slots >= 0  # the oracle, like the reference, has no notion of a token without a slot
    if keep.any():
        pa_oracle.reshape_and_cache(np.ascontiguousarray(inp["qkv"][keep, 1]), np.ascontiguousarray(inp["qkv"][keep, 2]),
                                    kref[0], vref[0], slots[keep], dtype)
    torch.cuda.synchronize()
    assert np.array_equal(datagen.torch_to_storage(kcs[0], dtype), kref[0]), what + ": key cache"
    assert np.array_equal(datagen.torch_to_storage(vcs[0], dtype), vref[0]), what + ": value cache"
    back = torch.zeros_like(qkv)
    pkg._capi.gather_cached_kv(back[:, 1], back[:, 2], kcs[0], vcs[0], slots_d)
    torch.cuda.synchronize()
    have = torch.from_numpy(slots >= 0).to(dev)
    assert torch.equal(back[have][:, 1:], qkv[have][:, 1:]) and not back[~have].any(), what + ": gather"
    if M:
        pkg._capi.copy_blocks(kcs, vcs, [int(x) for x in inp["src_blocks"]], [int(x) for x in inp["dst_blocks"]])
        pa_oracle.copy_blocks(kref, vref, inp["src_blocks"], inp["dst_blocks"])
        torch.cuda.synchronize()
        for l in range(L):
            assert np.array_equal(datagen.torch_to_storage(kcs[l], dtype), kref[l]), what + f": copy_blocks key layer {l}"
            assert np.array_equal(datagen.torch_to_storage(vcs[l], dtype), vref[l]), what + f": copy_blocks value layer {l}"


def rotary_cache_case(pkg, seed):
    """Random shape of rotary_embedding_neox + reshape_and_cache in one call (full and partial rotary,
    decode-sized and other batches, tokens sharing a block, tokens without a slot), bit for bit
    against the oracle's two steps."""
    import numpy as np
    from tests import test_rotary_cache_gpu as t
    rng = np.random.default_rng(11000 + seed)
    dtype = ["float16", "bfloat16", "float32"][seed % 3]
    D, B = int(rng.choice([64, 80, 96, 128])), int(rng.choice([8, 16, 32]))
    H = int(rng.choice([1, 2, 5, 8, 12]))
    T = int(rng.choice([1, 7, 16, 33, 100, 256, 511, 520]))
    x = 8 if dtype != "float32" else 4
    rot = int(rng.choice([D, D // 2, D // 4, 2 * x, 4 * x]))  # halves that are whole vectors; others are in the test suite
    NB = T + 3
    slots = None
    if rng.random() < 0.5:
        blocks = rng.permutation(NB)[:T]
        slots = (blocks * B + rng.integers(0, B, size=T)).astype(np.int32)
        if T > 2:
            slots[1] = slots[0] - 1 if slots[0] % B else slots[0] + 1   # two tokens in one block
            slots[T // 2] = -1                                           # a token without a slot
    t._check(pkg, dtype, T, H, D, B, rot, NB, seed=seed, slots=slots, via="capi" if seed % 2 else "module")


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    pkg = __graft_entry__.load_package()
    from tests import test_decode_gpu, test_prompt_attention
    failures = []
    t0 = time.time()
    for seed in range(first, first + count):
        for name, fn in (("decode", test_decode_gpu.test_decode_randomized_shapes_against_oracle),
                         ("prompt", test_prompt_attention.test_kernel_randomized_shapes_against_oracle),
                         ("append", append_case), ("cache", cache_case), ("rotary+cache", rotary_cache_case)):
            try:
                fn(pkg, seed)
            except AssertionError as e:
                failures.append((name, seed, str(e)[:300]))
                print(f"FAIL {name} seed {seed}: {str(e)[:300]}", flush=True)
        if (seed - first + 1) % 10 == 0:
            print(f"  {seed - first + 1} seeds, {len(failures)} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"seeds {first}..{first + count - 1}: decode, prompt attention, the append form, the cache operators and rotary + cache write against the oracle, {len(failures)} failures")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
