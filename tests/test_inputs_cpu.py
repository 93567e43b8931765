"""CPU tests of the input builder against a loop restatement of
Worker._prepare_inputs (vllm/worker/worker.py:128-237)."""
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import datagen
from oracle.inputs_oracle import prepare_inputs_lists


class SeqData:
    """The three accessors of vllm.sequence.SequenceData the builder uses."""

    def __init__(self, tokens):
        self.tokens = list(tokens)

    def get_len(self):
        return len(self.tokens)

    def get_token_ids(self):
        return self.tokens

    def get_last_token_id(self):
        return self.tokens[-1]


def make_batch(seed, num_prompt_groups, num_gen_groups, block_size, profiling=False):
    groups = []
    next_id, next_block = 0, 1
    r = datagen.randint(4 * (num_prompt_groups + num_gen_groups) + 8, seed, 1, 70)
    k = 0
    for _ in range(num_prompt_groups):
        n = int(r[k]); k += 1
        ids = [next_id, next_id + 1] if r[k] % 3 == 0 else [next_id]
        k += 1
        next_id += len(ids)
        toks = [int(t) for t in datagen.randint(n, seed + next_id, 0, 31999)]
        nb = (n + block_size - 1) // block_size
        table = list(range(next_block, next_block + nb))[::-1]
        next_block += nb
        groups.append(SimpleNamespace(
            request_id=str(next_id), is_prompt=True, seq_data={i: SeqData(toks) for i in ids},
            sampling_params=SimpleNamespace(n=len(ids)),
            block_tables=None if profiling else {i: table for i in ids}))
    for _ in range(num_gen_groups):
        nseq = 1 + int(r[k]) % 3; k += 1
        seq_data, tables = {}, {}
        for _s in range(nseq):
            n = int(r[k % len(r)]) + _s; k += 1
            toks = [int(t) for t in datagen.randint(n, seed + 100 + next_id, 0, 31999)]
            nb = (n + block_size - 1) // block_size
            tables[next_id] = list(range(next_block, next_block + nb))
            next_block += nb
            seq_data[next_id] = SeqData(toks)
            next_id += 1
        groups.append(SimpleNamespace(request_id=str(next_id), is_prompt=False, seq_data=seq_data,
                                      sampling_params=SimpleNamespace(n=nseq), block_tables=tables))
    # the scheduler may interleave prompt and generation groups
    order = datagen.permutation(len(groups), seed + 7)
    return [groups[i] for i in order]


@pytest.mark.parametrize("block_size", [8, 16, 32])
@pytest.mark.parametrize("shape", [(3, 5), (0, 7), (4, 0), (1, 1), (0, 0)])
def test_prepare_inputs_matches_reference_loops(pkg, block_size, shape):
    from vllm_amd.worker import prepare_inputs
    batch = make_batch(11 + block_size, shape[0], shape[1], block_size)
    want = prepare_inputs_lists(batch, block_size)
    tokens, positions, meta = prepare_inputs(batch, block_size, device="cpu")
    assert tokens.dtype.is_floating_point is False and str(tokens.dtype) == "torch.int64"
    assert tokens.tolist() == want["input_tokens"]
    assert positions.tolist() == want["input_positions"]
    assert str(meta.slot_mapping.dtype) == "torch.int32" and meta.slot_mapping.tolist() == want["slot_mapping"]
    assert str(meta.context_lens.dtype) == "torch.int32" and meta.context_lens.tolist() == want["context_lens"]
    assert str(meta.block_tables.dtype) == "torch.int32"
    assert meta.block_tables.tolist() == want["block_tables"]
    assert meta.prompt_lens == want["prompt_lens"]
    assert meta.max_context_len == want["max_context_len"]
    assert [ids for ids, _ in meta.seq_groups] == [ids for ids, _ in want["seq_groups"]]
    # derived attributes (vllm/model_executor/input_metadata.py:30-40)
    assert meta.num_prompts == len(want["prompt_lens"])
    assert meta.num_prompt_tokens == sum(want["prompt_lens"])
    assert meta.num_generation_tokens == len(want["context_lens"])
    assert meta.num_valid_tokens == len(want["slot_mapping"])
    assert meta.max_num_blocks_per_seq == (len(want["block_tables"][0]) if want["block_tables"] else 0)
    assert meta.attn_bias.seqlens == tuple(want["prompt_lens"])
    assert len(tokens) % 8 == 0 and len(tokens) - meta.num_valid_tokens < 8


def test_prepare_inputs_profiling_run_uses_slot_zero(pkg):
    from vllm_amd.worker import prepare_inputs
    batch = make_batch(5, 3, 0, 16, profiling=True)
    want = prepare_inputs_lists(batch, 16)
    _, _, meta = prepare_inputs(batch, 16, device="cpu")
    assert meta.slot_mapping.tolist() == want["slot_mapping"]
    assert set(meta.slot_mapping.tolist()) == {0}


def test_cache_block_size_formula(pkg):
    import torch
    from vllm_amd.worker import CacheEngine, simple_configs
    cc, mc, pc = simple_configs(num_layers=32, num_heads=32, head_size=128, dtype=torch.float16, block_size=16,
                                num_gpu_blocks=4, num_cpu_blocks=2)
    # vllm/worker/cache_engine.py:132-146: 2 * layers * block * heads * head_size * sizeof
    assert CacheEngine.get_cache_block_size(16, mc, pc) == 2 * 32 * 16 * 32 * 128 * 2
