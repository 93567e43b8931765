"""Carrier of the per-step tensors the attention layer consumes.

Mirror of vllm/model_executor/input_metadata.py:10-54 (same constructor
arguments, same derived attributes).  The one difference: `attn_bias` is a
plain description of the block-diagonal causal mask (the prompt lengths and
their prefix sums) instead of an xformers object: it only carries the
sequence lengths that attention_ops.multi_query_kv_attention (the MFMA prompt
kernel of this package, csrc/pa_prefill.hip) takes, both from
layers/attention.py and through the xformers import shim under compat/.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Tuple

import torch


@dataclass(frozen=True)
class BlockDiagonalCausal:
    """Each prompt attends causally to itself only (the mask xformers'
    BlockDiagonalCausalMask.from_seqlens encodes, input_metadata.py:30)."""
    seqlens: Tuple[int, ...]
    cu_seqlens: Tuple[int, ...]

    @staticmethod
    def from_seqlens(seqlens: List[int]) -> "BlockDiagonalCausal":
        cu = [0]
        for n in seqlens:
            cu.append(cu[-1] + int(n))
        return BlockDiagonalCausal(tuple(int(n) for n in seqlens), tuple(cu))


class InputMetadata:

    def __init__(
        self,
        seq_groups: List[Tuple[List[int], Any]],   # (seq_ids, sampling_params)
        seq_data: Dict[int, Any],                  # seq_id -> SequenceData
        prompt_lens: List[int],
        slot_mapping: torch.Tensor,                # [num_valid_tokens] int32
        context_lens: torch.Tensor,                # [num_generation_tokens] int32
        max_context_len: int,
        block_tables: torch.Tensor,                # [num_generation_tokens, max_blocks] int32
    ) -> None:
        self.seq_groups = seq_groups
        self.seq_data = seq_data
        self.prompt_lens = prompt_lens
        self.slot_mapping = slot_mapping
        self.context_lens = context_lens
        self.max_context_len = max_context_len
        self.block_tables = block_tables

        self.attn_bias = BlockDiagonalCausal.from_seqlens(prompt_lens)
        self.num_prompts = len(prompt_lens)
        self.num_prompt_tokens = sum(prompt_lens)
        self.num_generation_tokens = context_lens.shape[0]
        self.num_valid_tokens = slot_mapping.shape[0]
        if block_tables.numel() > 0:
            self.max_num_blocks_per_seq = block_tables.shape[1]
        else:
            self.max_num_blocks_per_seq = 0
        assert block_tables.shape[0] == self.num_generation_tokens
        assert context_lens.shape[0] == self.num_generation_tokens

    def __repr__(self) -> str:
        return (f"InputMetadata(num_valid_tokens={self.num_valid_tokens}, "
                f"num_prompt_tokens={self.num_prompt_tokens}, num_prompts={self.num_prompts}, "
                f"prompt_lens={self.prompt_lens}, num_generation_tokens={self.num_generation_tokens}, "
                f"max_context_len={self.max_context_len}, "
                f"max_num_blocks_per_seq={self.max_num_blocks_per_seq})")
