#!/usr/bin/env python3
"""Generates tests/golden/elementwise_golden.{npz,json} by running the
REFERENCE's own Python statements of rotary_embedding_neox, rms_norm and
silu_and_mul on CPU (build container only; reads /root/reference).

Taken from the reference, compiled from the files' syntax trees (the modules
themselves import CUDA extensions at the top):
  tests/kernels/test_pos_encoding.py:10-77   rotate_half, apply_rotary_pos_emb, RefRotaryEmbeddingNeox
  tests/kernels/test_layernorm.py:7-22       RefRMSNorm
  tests/kernels/test_activation.py:7-10      ref_silu_and_mul
Fixtures hold case parameters, an input checksum and the expected outputs.
"""
from __future__ import annotations

import ast
import json
import os
import sys
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import datagen  # noqa: E402

REF = "/root/reference/tests/kernels"
GOLDEN = os.path.join(ROOT, "tests", "golden")
TORCH_DTYPE = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}


def _load(path, names):
    tree = ast.parse(open(path, encoding="utf-8").read())
    nodes = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert len(nodes) == len(names), (names, [n.name for n in nodes])
    ns = {"torch": torch, "nn": nn, "F": F, "Tuple": Tuple}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, "exec"), ns)
    return ns


def main():
    rot = _load(os.path.join(REF, "test_pos_encoding.py"), ["rotate_half", "apply_rotary_pos_emb", "RefRotaryEmbeddingNeox"])
    rms = _load(os.path.join(REF, "test_layernorm.py"), ["RefRMSNorm"])
    act = _load(os.path.join(REF, "test_activation.py"), ["ref_silu_and_mul"])
    out, metas = {}, {}
    for case in datagen.elementwise_cases():
        inp = datagen.make_elementwise_inputs(case)
        to_t = lambda a: datagen.storage_to_torch(np.ascontiguousarray(a).copy(), case.dtype)
        with torch.inference_mode():
            if case.op == "rotary":
                T, H, D = case.num_tokens, case.num_heads, case.head_size
                # the reference module rebuilds cos/sin itself (test_pos_encoding.py:43-50); to pin the
                # kernel's arithmetic on the SAME table the op receives, its buffers are replaced by that table
                m = rot["RefRotaryEmbeddingNeox"](dim=case.rot_dim, max_position_embeddings=case.max_position)
                cache = to_t(inp["cos_sin_cache"])
                half = case.rot_dim // 2
                m.cos_cached = torch.cat((cache[:, :half], cache[:, :half]), dim=-1)
                m.sin_cached = torch.cat((cache[:, half:], cache[:, half:]), dim=-1)
                q, k = m(torch.from_numpy(inp["positions"]), to_t(inp["query"]).view(T, H, D), to_t(inp["key"]).view(T, H, D))
                out[case.name + "/query"] = datagen.torch_to_storage(q.reshape(T, H * D), case.dtype)
                out[case.name + "/key"] = datagen.torch_to_storage(k.reshape(T, H * D), case.dtype)
            elif case.op == "rms_norm":
                m = rms["RefRMSNorm"](case.hidden).to(TORCH_DTYPE[case.dtype])
                m.weight.data.copy_(to_t(inp["weight"]))
                out[case.name + "/out"] = datagen.torch_to_storage(m(to_t(inp["input"])), case.dtype)
            else:
                out[case.name + "/out"] = datagen.torch_to_storage(act["ref_silu_and_mul"](to_t(inp["input"])), case.dtype)
        metas[case.name] = {"case": json.loads(case.to_json()), "inputs_sha256": datagen.checksum_arrays(inp)}
        print("[elementwise]", case.name, flush=True)
    np.savez_compressed(os.path.join(GOLDEN, "elementwise_golden.npz"), **out)
    with open(os.path.join(GOLDEN, "elementwise_golden.json"), "w") as f:
        json.dump(metas, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
