#!/usr/bin/env python3
"""Longer differential run of the randomized parity tests: the same generators as
test_decode_randomized_shapes_against_oracle and test_kernel_randomized_shapes_against_oracle
(prompt attention), plus random shapes of the append form, on seeds beyond the ones the test suite
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
                         ("append", append_case)):
            try:
                fn(pkg, seed)
            except AssertionError as e:
                failures.append((name, seed, str(e)[:300]))
                print(f"FAIL {name} seed {seed}: {str(e)[:300]}", flush=True)
        if (seed - first + 1) % 10 == 0:
            print(f"  {seed - first + 1} seeds, {len(failures)} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"seeds {first}..{first + count - 1}: decode, prompt attention and the append form against the oracle, {len(failures)} failures")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
