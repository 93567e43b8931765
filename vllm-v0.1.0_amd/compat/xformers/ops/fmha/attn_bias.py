"""Mask descriptors.  BlockDiagonalCausalMask.from_seqlens(prompt_lens) is what
InputMetadata builds (vllm/model_executor/input_metadata.py:30): token i of a
prompt attends to tokens 0..i of the same prompt."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Sequence, Tuple


class AttentionBias:
    pass


@dataclass(frozen=True)
class BlockDiagonalCausalMask(AttentionBias):
    seqlens: Tuple[int, ...]

    @classmethod
    def from_seqlens(cls, q_seqlen: Sequence[int], kv_seqlen: Sequence[int] = None) -> "BlockDiagonalCausalMask":
        if kv_seqlen is not None and list(kv_seqlen) != list(q_seqlen):
            raise NotImplementedError("query and key/value lengths must match")
        if any(int(n) < 0 for n in q_seqlen):
            raise ValueError("negative sequence length")
        return cls(tuple(int(n) for n in q_seqlen))
