"""One tensor-parallel rank's share of a decode step's attention path.

The reference shards attention by heads: rank r of tp owns heads
[r * H / tp, (r + 1) * H / tp) (vllm/model_executor/models/llama.py:86-89,
opt.py:65-68, gpt_neox.py:51-53), its qkv projection produces only those heads
(llama.py:93-99), its CacheEngine holds only their K/V
(vllm/worker/cache_engine.py:31), block tables and context lengths are the same
on every rank (vllm/engine/llm_engine.py:214-220).  reshape_and_cache and
single_query_cached_kv_attention therefore run without any communication; the
exchange step is the all-reduce of the row-parallel o_proj that follows
(parallel_utils/tensor_parallel/layers.py:416-426).

Per layer, on the caller's stream unless noted:

  1. cache_ops.reshape_and_cache      new K/V of the local heads -> local cache
  2. attention_ops.single_query_cached_kv_attention[_planned]   local heads
  3. out_local [S, H/tp * D] @ w_o_local [H/tp * D, hidden] -> the launcher's
     next ring buffer (partial sums; waits for that buffer's previous reduce)
  4. launcher.launch_async(buffer)    RCCL all-reduce on the SIDE stream

and returns the Pending reduce.  The caller waits for it where the sum is
needed; whatever it queues before that (the next layer's steps 1-2 in the
hot-path benchmark) overlaps the collective.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .all_reduce import GraphAllReduce, Pending


def head_shard(rank: int, tp: int, num_heads: int) -> Tuple[int, int]:
    """[first, last) of the heads rank `rank` of `tp` owns; num_heads must divide evenly
    (the reference asserts the same, llama.py:86)."""
    if tp <= 0 or not 0 <= rank < tp:
        raise ValueError(f"rank {rank} is not in a tensor-parallel group of {tp}")
    if num_heads % tp != 0:
        raise ValueError(f"{num_heads} heads cannot be split evenly over {tp} tensor-parallel ranks")
    per = num_heads // tp
    return rank * per, (rank + 1) * per


class TensorParallelDecodeStep:

    def __init__(self, rank: int, tp: int, num_heads: int, head_size: int, scale: float,
                 launcher: Optional[GraphAllReduce], attention_ops=None, cache_ops=None) -> None:
        if attention_ops is None or cache_ops is None:
            from .. import attention_ops as _a, cache_ops as _c  # HIP modules; no fallback
            attention_ops = attention_ops or _a
            cache_ops = cache_ops or _c
        self._attention_ops = attention_ops
        self._cache_ops = cache_ops
        self.rank, self.tp = rank, tp
        self.first_head, self.last_head = head_shard(rank, tp, num_heads)
        self.local_heads = self.last_head - self.first_head
        self.head_size = head_size
        self.scale = float(scale)
        self.launcher = launcher
        if tp > 1 and launcher is None:
            raise ValueError("a tensor-parallel group of more than one rank needs an all-reduce launcher")
        self._workspace = None
        self._workspace_key = None

    def begin_step(self) -> None:
        """Call once per decode step (context_lens changed): the first layer re-plans."""
        self._workspace_key = None

    def _attend(self, out, query, key_cache, value_cache, block_tables, context_lens, max_context_len,
                total_context_len) -> None:
        block_size = value_cache.shape[3]
        planned = getattr(self._attention_ops, "single_query_cached_kv_attention_planned", None)
        if planned is None:
            self._attention_ops.single_query_cached_kv_attention(
                out, query, key_cache, value_cache, self.scale, block_tables, context_lens, block_size,
                max_context_len)
            return
        key = (query.shape[0], self.local_heads, self.head_size, block_size, max_context_len)
        ready = self._workspace_key == key
        if not ready:
            nbytes = self._attention_ops.decode_workspace_bytes(*key)
            if self._workspace is None or self._workspace.numel() < nbytes:
                self._workspace = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=query.device)
            self._workspace_key = key
        planned(out, query, key_cache, value_cache, self.scale, block_tables, context_lens, block_size,
                max_context_len, self._workspace, ready, total_context_len)

    def layer(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
              value_cache: torch.Tensor, slot_mapping: torch.Tensor, block_tables: torch.Tensor,
              context_lens: torch.Tensor, max_context_len: int, w_o_local: torch.Tensor, out_local: torch.Tensor,
              total_context_len: int = -1) -> Optional[Pending]:
        """query/key/value [S, H/tp, D] (column slices of the local fused qkv), caches of the local
        heads, w_o_local [H/tp * D, hidden] (this rank's rows of o_proj), out_local [S, H/tp, D]
        scratch.  Returns the Pending all-reduce of o_proj's output (its wait() gives [S, hidden]),
        or, with tp == 1, None after writing nothing beyond out_local."""
        S = query.shape[0]
        self._cache_ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping)
        self._attend(out_local, query, key_cache, value_cache, block_tables, context_lens, max_context_len,
                     total_context_len)
        if self.tp == 1:
            return None
        partial = self.launcher.next_buffer(S)  # fences this buffer's previous reduce
        torch.matmul(out_local.view(S, self.local_heads * self.head_size), w_o_local, out=partial)
        return self.launcher.launch_async(partial)
