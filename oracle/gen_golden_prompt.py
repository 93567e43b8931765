#!/usr/bin/env python3
"""Generates tests/golden/prompt_golden.* by running the REFERENCE's own
Python restatement of prompt attention on CPU.  Build container only (reads
/root/reference); tests read the committed fixtures.

ref_masked_attention and ref_multi_query_kv_attention
(/root/reference/tests/kernels/test_attention.py:14-37, 69-101) are compiled
from that file's syntax tree (the module cannot be imported: it needs xformers
and the CUDA extension).  The second function builds its mask on device
'cuda'; that one string constant is rewritten to 'cpu' in the syntax tree,
nothing else changes.  Each case is evaluated twice: in the storage dtype (as
the reference test does) and on the same stored values upcast to fp32.

The fixtures hold case parameters, an input checksum and expected outputs.
"""
from __future__ import annotations

import ast
import json
import os
import sys
from typing import List, Optional

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import datagen  # noqa: E402

REF = "/root/reference/tests/kernels/test_attention.py"
GOLDEN = os.path.join(ROOT, "tests", "golden")


class _CudaToCpu(ast.NodeTransformer):
    def visit_Constant(self, node):
        if node.value == "cuda":
            return ast.copy_location(ast.Constant("cpu"), node)
        return node


def load_reference_functions() -> dict:
    tree = ast.parse(open(REF, encoding="utf-8").read())
    names = ["ref_masked_attention", "ref_multi_query_kv_attention"]
    nodes = [_CudaToCpu().visit(n) for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(nodes) == 2
    ns = {"torch": torch, "Optional": Optional, "List": List}
    exec(compile(ast.fix_missing_locations(ast.Module(body=nodes, type_ignores=[])), REF, "exec"), ns)
    return ns


def main() -> None:
    ref = load_reference_functions()["ref_multi_query_kv_attention"]
    out, metas = {}, {}
    for case in datagen.prompt_cases():
        inp = datagen.make_prompt_inputs(case)
        qkv = datagen.storage_to_torch(inp["qkv"], case.dtype)
        q, k, v = qkv.unbind(dim=1)
        cu = [int(x) for x in inp["cu_seqlens"]]
        exp_dtype = ref(cu, q, k, v, q.dtype)
        exp_f32 = ref(cu, q.float(), k.float(), v.float(), torch.float32)
        out[case.name + "/ref_dtype"] = datagen.torch_to_storage(exp_dtype.contiguous(), case.dtype)
        out[case.name + "/ref_f32"] = exp_f32.numpy().copy()
        metas[case.name] = {"case": json.loads(case.to_json()), "inputs_sha256": datagen.checksum_arrays(inp)}
        print(f"[prompt] {case.name} {tuple(exp_f32.shape)}", flush=True)
    np.savez_compressed(os.path.join(GOLDEN, "prompt_golden.npz"), **out)
    with open(os.path.join(GOLDEN, "prompt_golden.json"), "w") as f:
        json.dump(metas, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
