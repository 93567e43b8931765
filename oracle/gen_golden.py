#!/usr/bin/env python3
"""Generates tests/golden/* by running the REFERENCE's own Python statements
of the path on CPU.  Run in the build container only (it reads
/root/reference); the GPU box and the tests only read the committed fixtures.

What is taken from the reference, and how:
  * decode attention: the two functions ref_masked_attention and
    ref_single_query_cached_kv_attention
    (/root/reference/tests/kernels/test_attention.py:14-66) are compiled from
    that file's syntax tree (the module itself cannot be imported: it imports
    xformers and the CUDA extension at the top) and executed on CPU tensors.
  * cache ops: the inline reference loops of
    /root/reference/tests/kernels/test_cache.py (run_copy_blocks :51-56,
    run_reshape_and_cache :95-100, run_gather_cached_kv :138-143) are compiled
    from the syntax tree and executed on CPU tensors bound to the names the
    loops use.

No reference source text is written to the fixtures: they hold the case
parameters, a checksum of the (deterministically regenerated) inputs and the
expected outputs or their SHA-256 digests.
"""
from __future__ import annotations

import ast
import hashlib
import json
import os
import sys
import time
from typing import List, Optional

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import datagen  # noqa: E402
from oracle.datagen import CacheCase, DecodeCase  # noqa: E402

REF = "/root/reference/tests/kernels"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _load_functions(path: str, names: List[str]) -> dict:
    tree = ast.parse(open(path, encoding="utf-8").read())
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(nodes) == len(names), (names, [n.name for n in nodes])
    ns = {"torch": torch, "Optional": Optional, "List": List}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, "exec"), ns)
    return ns


def _load_reference_loop(path: str, func: str, target_hint: str):
    """Returns the compiled first `for` statement of `func` whose source
    mentions target_hint."""
    src = open(path, encoding="utf-8").read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == func)
    for node in fn.body:
        if isinstance(node, ast.For) and target_hint in ast.get_source_segment(src, node):
            return compile(ast.Module(body=[node], type_ignores=[]), f"{path}:{func}", "exec")
    raise RuntimeError(f"no reference loop found in {func}")


TORCH_DTYPE = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}


def decode_cases() -> List[DecodeCase]:
    cases: List[DecodeCase] = []
    seed = 1
    # the reference test's grid (test_attention.py:274-276) at small sizes,
    # once with its own input distribution and once at unit scale
    for dtype in datagen.DTYPES:
        for B in (8, 16, 32):
            for D in (64, 80, 96, 128):
                for dist in ("uniform1e-3", "normal"):
                    cases.append(DecodeCase(
                        name=f"grid_{dtype}_b{B}_d{D}_{dist}", dtype=dtype, num_seqs=4,
                        num_heads=3, head_size=D, block_size=B, num_blocks=24,
                        max_context=5 * B + 3, dist=dist, seed=seed))
                    seed += 1
    # edge cases: one token, exact multiples of the block, one past a block,
    # the longest at the front and at the back, padded tables
    for dtype in datagen.DTYPES:
        cases.append(DecodeCase(
            name=f"edges_{dtype}", dtype=dtype, num_seqs=8, num_heads=2, head_size=128,
            block_size=16, num_blocks=40, context_lens=[1, 16, 17, 15, 32, 33, 160, 2],
            dist="normal", table_pad=3, seed=seed))
        seed += 1
        cases.append(DecodeCase(
            name=f"onehead_{dtype}", dtype=dtype, num_seqs=3, num_heads=1, head_size=64,
            block_size=8, num_blocks=16, context_lens=[8, 1, 9], dist="normal",
            tables="permutation", seed=seed))
        seed += 1
    # the reference test's own sizes (test_attention.py:280-287): 37 tokens,
    # 3 heads, 1024 blocks, context up to 4096, one head size per
    # (dtype, block size)
    heads = (64, 80, 96, 128)
    k = 0
    for dtype in datagen.DTYPES:
        for B in (8, 16, 32):
            cases.append(DecodeCase(
                name=f"reftest_{dtype}_b{B}_d{heads[k % 4]}", dtype=dtype, num_seqs=37,
                num_heads=3, head_size=heads[k % 4], block_size=B, num_blocks=1024,
                max_context=4096, dist="uniform1e-3", seed=seed))
            seed += 1
            k += 1
    return cases


def decode_shard_cases() -> List[DecodeCase]:
    """Per-GPU shapes of the tensor-parallel configurations: LLaMA-13B over two
    GPUs = 20 heads x 128 per rank, GPT-NeoX-20B over four = 16 heads x 96 per
    rank with three samples per prompt (heads per rank:
    /root/reference/vllm/model_executor/models/llama.py:86-89, gpt_neox.py:51-53;
    forked block tables: /root/reference/vllm/core/block_manager.py:111-135), and
    the 12 heads x 64 of OPT-125m."""
    cases: List[DecodeCase] = []
    seed = 700
    for dtype in datagen.DTYPES:
        for dist in ("uniform1e-3", "normal"):
            cases.append(DecodeCase(
                name=f"llama13b_tp2_shard_{dtype}_{dist}", dtype=dtype, num_seqs=10, num_heads=20,
                head_size=128, block_size=16, num_blocks=160, max_context=230, dist=dist,
                tables="permutation", seed=seed))
            seed += 1
            cases.append(DecodeCase(
                name=f"neox20b_tp4_shard_n3_{dtype}_{dist}", dtype=dtype, num_seqs=12, num_heads=16,
                head_size=96, block_size=16, num_blocks=200, max_context=400, dist=dist,
                tables="forked", fork=3, seed=seed))
            seed += 1
        cases.append(DecodeCase(
            name=f"opt125m_{dtype}", dtype=dtype, num_seqs=9, num_heads=12, head_size=64,
            block_size=16, num_blocks=96, max_context=150, dist="normal", tables="permutation", seed=seed))
        seed += 1
    # forked groups with unequal lengths inside a group, a group cut short by the end of
    # the batch, other block sizes
    for B, D in ((8, 96), (32, 80)):
        cases.append(DecodeCase(
            name=f"forked_b{B}_d{D}", dtype="float16", num_seqs=8, num_heads=4, head_size=D,
            block_size=B, num_blocks=120, context_lens=[70, 75, 66, 200, 203, 200, 9, 12],
            dist="normal", tables="forked", fork=3, seed=seed))
        seed += 1
    return cases


def gen_decode(ref_ns: dict, cases=None, stem: str = "decode_golden") -> None:
    ref_fn = ref_ns["ref_single_query_cached_kv_attention"]
    out = {}
    metas = {}
    t0 = time.time()
    for case in (decode_cases() if cases is None else cases):
        inp = datagen.make_decode_inputs(case)
        q = datagen.storage_to_torch(inp["qkv"], case.dtype)[:, 0]
        kc = datagen.storage_to_torch(inp["key_cache"], case.dtype)
        vc = datagen.storage_to_torch(inp["value_cache"], case.dtype)
        bt = torch.from_numpy(inp["block_tables"])
        cl = torch.from_numpy(inp["context_lens"])
        # (1) exactly as the reference test runs it: in the storage dtype
        exp_dtype = torch.empty(q.shape, dtype=q.dtype)
        ref_fn(exp_dtype, q, kc, vc, bt, cl)
        # (2) the same function on the same stored values upcast to fp32:
        #     the reference formula without storage-type arithmetic noise
        exp_f32 = torch.empty(q.shape, dtype=torch.float32)
        ref_fn(exp_f32, q.float(), kc.float(), vc.float(), bt, cl)
        out[case.name + "/ref_dtype"] = datagen.torch_to_storage(exp_dtype, case.dtype)
        out[case.name + "/ref_f32"] = exp_f32.numpy().copy()
        metas[case.name] = {"case": json.loads(case.to_json()),
                            "inputs_sha256": datagen.checksum(inp)}
        print(f"[decode] {case.name}  ({time.time() - t0:.1f}s)", flush=True)
    np.savez_compressed(os.path.join(GOLDEN, stem + ".npz"), **out)
    with open(os.path.join(GOLDEN, stem + ".json"), "w") as f:
        json.dump(metas, f, indent=1, sort_keys=True)


def _digest(arrs) -> str:
    h = hashlib.sha256()
    for a in arrs:
        a = np.ascontiguousarray(a)
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def cache_cases() -> List[CacheCase]:
    cases: List[CacheCase] = []
    seed = 100
    for dtype in datagen.DTYPES:
        # the reference test's sizes (test_cache.py:149-170)
        cases.append(CacheCase(name=f"reshape_reftest_{dtype}", dtype=dtype, num_tokens=3,
                               num_heads=2, head_size=16, block_size=8, num_blocks=2, seed=seed))
        cases.append(CacheCase(name=f"copy_reftest_{dtype}", dtype=dtype, num_tokens=0,
                               num_heads=17, head_size=16, block_size=8, num_blocks=1024,
                               num_layers=7, num_mappings=23, seed=seed + 1))
        # decode-step shaped and odd shapes
        cases.append(CacheCase(name=f"reshape_decode_{dtype}", dtype=dtype, num_tokens=37,
                               num_heads=5, head_size=128, block_size=16, num_blocks=19,
                               seed=seed + 2))
        cases.append(CacheCase(name=f"reshape_d80_b32_{dtype}", dtype=dtype, num_tokens=11,
                               num_heads=3, head_size=80, block_size=32, num_blocks=4,
                               seed=seed + 3))
        cases.append(CacheCase(name=f"copy_small_{dtype}", dtype=dtype, num_tokens=0,
                               num_heads=3, head_size=64, block_size=16, num_blocks=12,
                               num_layers=2, num_mappings=5, seed=seed + 4))
        seed += 10
    return cases


def gen_cache() -> None:
    path = os.path.join(REF, "test_cache.py")
    loop_copy = _load_reference_loop(path, "run_copy_blocks", "block_mapping.items()")
    loop_reshape = _load_reference_loop(path, "run_reshape_and_cache", "range(num_tokens)")
    loop_gather = _load_reference_loop(path, "run_gather_cached_kv", "range(num_tokens)")
    metas = {}
    full = {}
    for case in cache_cases():
        inp = datagen.make_cache_inputs(case)
        T, H, D, B = case.num_tokens, case.num_heads, case.head_size, case.block_size
        x = 16 // datagen.ELEM_SIZE[case.dtype]
        entry = {"case": json.loads(case.to_json())}
        to_t = lambda a: datagen.storage_to_torch(np.ascontiguousarray(a).copy(), case.dtype)
        if case.name.startswith("copy"):
            key_caches = [to_t(a) for a in inp["key_caches"]]
            value_caches = [to_t(a) for a in inp["value_caches"]]
            ns = {
                "block_mapping": {int(s): [int(d)] for s, d in zip(inp["src_blocks"], inp["dst_blocks"])},
                "key_caches": key_caches, "value_caches": value_caches,
                "cloned_key_caches": key_caches, "cloned_value_caches": value_caches,
                "zip": zip,
            }
            exec(loop_copy, ns)
            entry["key_caches_sha256"] = _digest(datagen.torch_to_storage(t, case.dtype) for t in key_caches)
            entry["value_caches_sha256"] = _digest(datagen.torch_to_storage(t, case.dtype) for t in value_caches)
        else:
            qkv = to_t(inp["qkv"])
            _, key, value = qkv.unbind(dim=1)
            kc0, vc0 = to_t(inp["key_caches"][0]), to_t(inp["value_caches"][0])
            slot_mapping = torch.from_numpy(inp["slot_mapping"].copy())
            # reshape_and_cache (:95-100)
            ns = {"torch": torch, "num_tokens": T, "num_heads": H, "head_size": D, "x": x,
                  "block_size": B, "key": key, "value": value, "slot_mapping": slot_mapping,
                  "cloned_key_cache": kc0, "cloned_value_cache": vc0, "range": range}
            exec(loop_reshape, ns)
            kc_np = datagen.torch_to_storage(kc0, case.dtype)
            vc_np = datagen.torch_to_storage(vc0, case.dtype)
            entry["reshape_key_cache_sha256"] = _digest([kc_np])
            entry["reshape_value_cache_sha256"] = _digest([vc_np])
            # gather_cached_kv (:138-143) from the ORIGINAL caches into a
            # fresh qkv clone
            qkv2 = to_t(inp["qkv"])
            _, cloned_key, cloned_value = qkv2.unbind(dim=1)
            ns = {"torch": torch, "num_tokens": T, "num_heads": H, "head_size": D, "x": x,
                  "block_size": B, "cloned_key": cloned_key, "cloned_value": cloned_value,
                  "slot_mapping": slot_mapping, "key_cache": to_t(inp["key_caches"][0]),
                  "value_cache": to_t(inp["value_caches"][0]), "range": range}
            exec(loop_gather, ns)
            qkv2_np = datagen.torch_to_storage(qkv2, case.dtype)
            entry["gather_qkv_sha256"] = _digest([qkv2_np])
            if "reftest" in case.name:  # tiny: keep the full expected arrays too
                full[case.name + "/reshape_key_cache"] = kc_np
                full[case.name + "/reshape_value_cache"] = vc_np
                full[case.name + "/gather_qkv"] = qkv2_np
        metas[case.name] = entry
        print(f"[cache] {case.name}", flush=True)
    with open(os.path.join(GOLDEN, "cache_golden.json"), "w") as f:
        json.dump(metas, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(GOLDEN, "cache_golden_small.npz"), **full)


def main() -> None:
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    ref_ns = _load_functions(os.path.join(REF, "test_attention.py"),
                             ["ref_masked_attention", "ref_single_query_cached_kv_attention"])
    only = sys.argv[1] if len(sys.argv) > 1 else "all"  # "all" | "shards"
    if only == "all":
        gen_cache()
        gen_decode(ref_ns)
    gen_decode(ref_ns, decode_shard_cases(), "decode_golden_shards")


if __name__ == "__main__":
    main()
