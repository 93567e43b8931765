"""Attention layer over the paged KV cache.

Counterpart of PagedAttention (vllm/model_executor/layers/attention.py:16-156):
same constructor, same `forward` arguments, same treatment of the flat token
batch

    |<-------------------- num_valid_tokens -------------------->|
    |<----- num_prompt_tokens ----->|<-- num_generation_tokens -->|<-padding->|

  1. prompt tokens: causal self-attention per prompt, no cache read (:115-122);
  2. wait for the cache engine's swap of this layer (:125-126);
  3. reshape_and_cache on [:num_valid_tokens] (:131-140), here as one call for the
     prompt tokens and one for the generation tokens;
  4. single_query_cached_kv_attention on the generation tokens (:142-152);
     its launch plan is shared by the layers of a step (see below).
     For small generation batches (at most _APPEND_FORM_MAX_ROWS sequence-head rows) whose
     slots are known to follow the block tables, steps 3 and 4 of the generation tokens are one
     launch: the attention kernel puts the new token into the cache itself.

Steps 1, 3 and 4 are the HIP operator modules.  For step 1 the reference calls
xformers' CUTLASS kernel with a block-diagonal causal mask (:59-67); here it is
one matrix-core kernel launch over all prompts of the batch.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ..input_metadata import BlockDiagonalCausal, InputMetadata

_SUPPORTED_HEAD_SIZES = [64, 80, 96, 128]
# Largest generation batch, in (sequence, head) rows, whose cache write rides in the attention
# launch.  Measured on MI355X, 32 heads x 128, 1024 tokens: 1 sequence 18.4 -> 11.5 us per layer,
# 16 sequences 48.1 -> 45.5; from 32 sequences on the separate reshape_and_cache launch is faster
# (stores scattered through a launch that saturates HBM with reads are expensive).
_APPEND_FORM_MAX_ROWS = 640


class PagedAttention(nn.Module):

    def __init__(self, num_heads: int, head_size: int, scale: float, attention_ops=None, cache_ops=None) -> None:
        super().__init__()
        if attention_ops is None or cache_ops is None:
            from .. import attention_ops as _a, cache_ops as _c  # HIP modules; no fallback
            attention_ops = attention_ops or _a
            cache_ops = cache_ops or _c
        self._attention_ops = attention_ops
        self._cache_ops = cache_ops
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        if self.head_size not in _SUPPORTED_HEAD_SIZES:
            raise ValueError(f"head_size ({self.head_size}) is not supported. "
                             f"Supported head sizes: {_SUPPORTED_HEAD_SIZES}.")

    def multi_query_kv_attention(
        self,
        output: torch.Tensor,   # [num_prompt_tokens, num_heads, head_size]
        query: torch.Tensor,    # [num_prompt_tokens, num_heads, head_size]
        key: torch.Tensor,
        value: torch.Tensor,
        attn_bias: BlockDiagonalCausal,
    ) -> torch.Tensor:
        self._attention_ops.multi_query_kv_attention(output, query, key, value, list(attn_bias.seqlens), self.scale)
        return output

    def single_query_cached_kv_attention(
        self,
        output: torch.Tensor,       # [num_generation_tokens, num_heads, head_size]
        query: torch.Tensor,        # [num_generation_tokens, num_heads, head_size]
        key_cache: torch.Tensor,    # [num_blocks, num_heads, head_size/x, block_size, x]
        value_cache: torch.Tensor,  # [num_blocks, num_heads, head_size, block_size]
        input_metadata: InputMetadata,
    ) -> None:
        self._decode(output, query, None, None, key_cache, value_cache, input_metadata)

    def _decode(self, output, query, new_key, new_value, key_cache, value_cache, input_metadata) -> None:
        """new_key / new_value given: the generation tokens' cache write is part of the launch."""
        block_size = value_cache.shape[3]
        planned = getattr(self._attention_ops, "single_query_cached_kv_attention_planned", None)
        if planned is None:  # a module with the reference's functions only
            assert new_key is None
            self._attention_ops.single_query_cached_kv_attention(
                output, query, key_cache, value_cache, self.scale, input_metadata.block_tables,
                input_metadata.context_lens, block_size, input_metadata.max_context_len)
            return
        # Every layer of a step sees the same context_lens: the launch plan that
        # depends on them alone (row order, partitions of long contexts) is made
        # by the first layer into a workspace that lives on the step's
        # InputMetadata, and reused by the others.
        key = (query.shape[0], self.num_heads, self.head_size, block_size, input_metadata.max_context_len,
               query.device)
        cached = getattr(input_metadata, "_decode_workspace", None)
        plan_ready = cached is not None and cached[0] == key
        if not plan_ready:
            nbytes = self._attention_ops.decode_workspace_bytes(*key[:5])
            cached = (key, torch.empty(nbytes, dtype=torch.uint8, device=query.device))
            input_metadata._decode_workspace = cached
        total = int(getattr(input_metadata, "total_context_len", -1))
        if new_key is not None:
            self._attention_ops.reshape_and_cache_single_query_cached_kv_attention(
                output, query, new_key, new_value, key_cache, value_cache, self.scale, input_metadata.block_tables,
                input_metadata.context_lens, block_size, input_metadata.max_context_len, cached[1], plan_ready, total)
            return
        planned(output, query, key_cache, value_cache, self.scale, input_metadata.block_tables,
                input_metadata.context_lens, block_size, input_metadata.max_context_len, cached[1], plan_ready, total)

    def forward(
        self,
        query: torch.Tensor,                  # [num_tokens, num_heads * head_size]
        key: torch.Tensor,                    # [num_tokens, num_heads * head_size]
        value: torch.Tensor,                  # [num_tokens, num_heads * head_size]
        key_cache: Optional[torch.Tensor],
        value_cache: Optional[torch.Tensor],
        input_metadata: InputMetadata,
        cache_event: Optional[torch.cuda.Event],
        _generation_tokens_cached: bool = False,
    ) -> torch.Tensor:                        # [num_tokens, num_heads * head_size]
        # _generation_tokens_cached: the caller (PagedAttentionWithRoPE) has already written the
        # generation tokens' key and value into the caches.
        # query/key/value are column slices of the fused qkv projection: the
        # views below keep their row stride.
        query = query.view(-1, self.num_heads, self.head_size)
        key = key.view(-1, self.num_heads, self.head_size)
        value = value.view(-1, self.num_heads, self.head_size)
        output = torch.empty(query.shape, dtype=query.dtype, device=query.device)

        num_prompt_tokens = input_metadata.num_prompt_tokens
        if num_prompt_tokens > 0:
            self.multi_query_kv_attention(
                output[:num_prompt_tokens],
                query[:num_prompt_tokens],
                key[:num_prompt_tokens],
                value[:num_prompt_tokens],
                input_metadata.attn_bias,
            )

        if cache_event is not None:
            cache_event.wait()

        num_valid_tokens = input_metadata.num_valid_tokens
        num_generation_tokens = input_metadata.num_generation_tokens
        append_form = (
            not _generation_tokens_cached and key_cache is not None and value_cache is not None
            and 0 < num_generation_tokens * self.num_heads <= _APPEND_FORM_MAX_ROWS
            and num_valid_tokens == num_prompt_tokens + num_generation_tokens
            and getattr(input_metadata, "generation_slots_follow_tables", False)
            and hasattr(self._attention_ops, "reshape_and_cache_single_query_cached_kv_attention"))
        if num_valid_tokens > 0 and key_cache is not None and value_cache is not None:
            # One call per kind of token (the reference makes one call for all, :131-140):
            # prompts fill runs of consecutive slots, generation tokens one slot in a block
            # of their own, and the operator picks its kernel by what a call contains.
            spans = [(0, num_prompt_tokens)]
            if not _generation_tokens_cached and not append_form:
                spans.append((num_prompt_tokens, num_valid_tokens))
            for lo, hi in spans:
                if hi > lo:
                    self._cache_ops.reshape_and_cache(
                        key[lo:hi],
                        value[lo:hi],
                        key_cache,
                        value_cache,
                        input_metadata.slot_mapping[lo:hi],
                    )

        if input_metadata.num_generation_tokens > 0:
            assert key_cache is not None and value_cache is not None, (
                "key_cache and value_cache must be provided when generating tokens.")
            gen = slice(num_prompt_tokens, num_valid_tokens)
            if append_form:
                self._decode(output[gen], query[gen], key[gen], value[gen], key_cache, value_cache, input_metadata)
            else:
                self.single_query_cached_kv_attention(output[gen], query[gen], key_cache, value_cache, input_metadata)

        # the output may contain padding rows (left uninitialised, as in the reference)
        return output.view(-1, self.num_heads * self.head_size)


class PagedAttentionWithRoPE(PagedAttention):
    """PagedAttention preceded by the GPT-NeoX rotary embedding of query and key
    (vllm/model_executor/layers/attention.py:159-215): same constructor, same
    cos/sin table ([max_position, rotary_dim], cos half then sin half, in the
    default dtype), same `forward` arguments."""

    def __init__(self, num_heads: int, head_size: int, scale: float, rotary_dim: int,
                 max_position: int = 8192, base: int = 10000, attention_ops=None, cache_ops=None,
                 pos_encoding_ops=None) -> None:
        super().__init__(num_heads, head_size, scale, attention_ops, cache_ops)
        if pos_encoding_ops is None:
            from .. import pos_encoding_ops  # HIP module; no fallback
        self._pos_encoding_ops = pos_encoding_ops
        inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2) / rotary_dim))
        freqs = torch.outer(torch.arange(max_position).float(), inv_freq.float())
        table = torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(torch.get_default_dtype())
        self.register_buffer("cos_sin_cache", table, persistent=False)

    def forward(
        self,
        positions: torch.Tensor,              # [num_tokens]
        query: torch.Tensor,                  # [num_tokens, num_heads * head_size]
        key: torch.Tensor,
        value: torch.Tensor,
        key_cache: Optional[torch.Tensor],
        value_cache: Optional[torch.Tensor],
        input_metadata: InputMetadata,
        cache_event: Optional[torch.cuda.Event],
    ) -> torch.Tensor:
        # in place on the q and k column slices of the fused qkv row
        fused = getattr(self._cache_ops, "rotary_embedding_neox_reshape_and_cache", None)
        n_prompt, n_valid = input_metadata.num_prompt_tokens, input_metadata.num_valid_tokens
        if fused is None or key_cache is None or value_cache is None or n_valid <= n_prompt:
            self._pos_encoding_ops.rotary_embedding_neox(positions, query, key, self.head_size, self.cos_sin_cache)
            return super().forward(query, key, value, key_cache, value_cache, input_metadata, cache_event)
        # Generation tokens: rotation and cache write in one pass over the key (the rotated key of
        # a generation token has no other reader; prompt tokens' keys are also read by the prompt
        # attention, so they are rotated first and cached by the base class as before).  Padding
        # rows behind the valid tokens are left alone: nothing reads them.
        if n_prompt > 0:
            self._pos_encoding_ops.rotary_embedding_neox(positions[:n_prompt], query[:n_prompt], key[:n_prompt],
                                                         self.head_size, self.cos_sin_cache)
        if cache_event is not None:
            cache_event.wait()  # the swap of this layer's blocks must have landed before the cache is written
            cache_event = None
        H, D = self.num_heads, self.head_size
        fused(positions[n_prompt:n_valid], query[n_prompt:n_valid].view(-1, H, D), key[n_prompt:n_valid].view(-1, H, D),
              value[n_prompt:n_valid].view(-1, H, D), key_cache, value_cache,
              input_metadata.slot_mapping[n_prompt:n_valid], self.cos_sin_cache)
        return super().forward(query, key, value, key_cache, value_cache, input_metadata, cache_event,
                               _generation_tokens_cached=True)
