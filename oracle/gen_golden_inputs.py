#!/usr/bin/env python3
"""Generates tests/golden/inputs_golden.json by running the REFERENCE's own
Worker._prepare_inputs on CPU.  Run in the build container only (it reads
/root/reference); the tests only read the committed fixture.

How: the method `_prepare_inputs` of class Worker and the module functions
`_pad_to_alignment` / `_pad_to_max` (/root/reference/vllm/worker/worker.py:128-237,
304-309) are taken from the file's syntax tree (the module itself imports the
CUDA-only parts of vllm at the top).  The five `torch.cuda.LongTensor` /
`torch.cuda.IntTensor` constructors (:215-222) are rewritten in the tree to
their CPU counterparts `torch.LongTensor` / `torch.IntTensor`; nothing else is
changed.  `InputMetadata` is bound to a recorder of the keyword arguments the
method passes (the reference class builds an xformers mask, which is not part
of the builder's result).  The scheduler output is synthesised by
oracle/datagen.make_input_batch, so the fixture stores only the case
parameters and the expected lists — no reference source text.
"""
from __future__ import annotations

import ast
import json
import os
import sys
from types import SimpleNamespace
from typing import Dict, List, Tuple

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import datagen  # noqa: E402

REF_WORKER = "/root/reference/vllm/worker/worker.py"
GOLDEN = os.path.join(ROOT, "tests", "golden")


class _CudaCtorToCpu(ast.NodeTransformer):
    """torch.cuda.XTensor -> torch.XTensor"""

    def __init__(self):
        self.rewritten = 0

    def visit_Attribute(self, node):
        self.generic_visit(node)
        v = node.value
        if (node.attr in ("LongTensor", "IntTensor") and isinstance(v, ast.Attribute) and v.attr == "cuda"
                and isinstance(v.value, ast.Name) and v.value.id == "torch"):
            self.rewritten += 1
            return ast.copy_location(ast.Attribute(value=v.value, attr=node.attr, ctx=node.ctx), node)
        return node


def load_reference_prepare_inputs():
    tree = ast.parse(open(REF_WORKER, encoding="utf-8").read())
    worker = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Worker")
    method = next(n for n in worker.body if isinstance(n, ast.FunctionDef) and n.name == "_prepare_inputs")
    pads = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("_pad_to_alignment", "_pad_to_max")]
    assert len(pads) == 2
    rewriter = _CudaCtorToCpu()
    method = rewriter.visit(method)
    assert rewriter.rewritten == 5, rewriter.rewritten
    mod = ast.fix_missing_locations(ast.Module(body=pads + [method], type_ignores=[]))
    names = ("SequenceGroupMetadata", "SamplingParams", "SequenceData")
    ns = {"torch": torch, "List": List, "Tuple": Tuple, "Dict": Dict,
          "InputMetadata": lambda **kw: SimpleNamespace(**kw), **{n: object for n in names}}
    exec(compile(mod, REF_WORKER, "exec"), ns)
    return ns["_prepare_inputs"]


def main() -> None:
    ref = load_reference_prepare_inputs()
    out = []
    for case in datagen.input_batch_cases():
        batch = datagen.make_input_batch(case["seed"], case["prompts"], case["gens"], case["block_size"],
                                         case["profiling"])
        worker = SimpleNamespace(block_size=case["block_size"])
        tokens, positions, meta = ref(worker, batch)
        assert tokens.dtype == torch.int64 and positions.dtype == torch.int64
        assert meta.slot_mapping.dtype == torch.int32 and meta.context_lens.dtype == torch.int32
        assert meta.block_tables.dtype == torch.int32
        out.append({
            "case": case,
            "input_tokens": tokens.tolist(),
            "input_positions": positions.tolist(),
            "slot_mapping": meta.slot_mapping.tolist(),
            "context_lens": meta.context_lens.tolist(),
            "block_tables": meta.block_tables.tolist(),
            "block_tables_shape": list(meta.block_tables.shape),
            "prompt_lens": list(meta.prompt_lens),
            "max_context_len": int(meta.max_context_len),
            "seq_groups": [list(ids) for ids, _ in meta.seq_groups],
            "seq_data_keys": sorted(meta.seq_data.keys()),
        })
        print(f"[inputs] {case}", flush=True)
    with open(os.path.join(GOLDEN, "inputs_golden.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
