"""Builds the per-step device tensors of the decode path from the scheduler's
output: token ids, positions, slot mapping, context lengths, block tables.

Counterpart of Worker._prepare_inputs (vllm/worker/worker.py:128-237) with the
same conventions:

  * batch order: all prompt groups first, then every sequence of every
    generation group (:139-207);
  * slot = block_table[pos // B] * B + pos % B (:166-172, :204-207); during
    memory profiling (`block_tables is None`) prompts get slot 0 (:160-164);
  * context_len = sequence length including the token being decoded (:192),
    position = context_len - 1 (:193);
  * tokens / positions padded with 0 to a multiple of 8 (:209-212), the slot
    mapping is NOT padded, block tables are right-padded with 0 to the widest
    one (:219-222, :308-309);
  * int64 tokens/positions, int32 everything else (:215-222).

The reference builds five Python lists element by element and makes five
separate synchronous host-to-device copies.  Here the index arithmetic is
vectorised (numpy), all arrays are packed into ONE pinned staging buffer and
sent with ONE asynchronous copy on the current stream; the returned tensors
are typed views of the single device buffer.
"""
from __future__ import annotations

from typing import Any, Dict, List, Sequence, Tuple

import numpy as np
import torch

from ..input_metadata import InputMetadata


def _pad_len(n: int, multiple_of: int) -> int:
    return n + (-n) % multiple_of


def _align(n: int, a: int = 16) -> int:
    return (n + a - 1) // a * a


def prepare_inputs(
    seq_group_metadata_list: Sequence[Any],
    block_size: int,
    device: torch.device | str = "cuda",
    pad_multiple: int = 8,
) -> Tuple[torch.Tensor, torch.Tensor, InputMetadata]:
    """seq_group_metadata_list: objects with the attributes of
    vllm.sequence.SequenceGroupMetadata (is_prompt, seq_data, sampling_params,
    block_tables); seq_data values need get_token_ids(), get_len() and
    get_last_token_id()."""
    B = int(block_size)
    seq_groups: List[Tuple[List[int], Any]] = []
    prompt_lens: List[int] = []
    tok_parts: List[np.ndarray] = []
    pos_parts: List[np.ndarray] = []
    slot_parts: List[np.ndarray] = []

    # prompts
    for g in seq_group_metadata_list:
        if not g.is_prompt:
            continue
        seq_ids = list(g.seq_data.keys())
        seq_groups.append((seq_ids, g.sampling_params))
        seq_id = seq_ids[0]  # any sequence of the group: they share the prompt
        toks = np.asarray(g.seq_data[seq_id].get_token_ids(), dtype=np.int64)
        n = int(toks.shape[0])
        prompt_lens.append(n)
        idx = np.arange(n, dtype=np.int64)
        tok_parts.append(toks)
        pos_parts.append(idx)
        if g.block_tables is None:
            slot_parts.append(np.zeros(n, dtype=np.int64))
        else:
            table = np.asarray(g.block_tables[seq_id], dtype=np.int64)
            slot_parts.append(table[idx // B] * B + idx % B)

    # generation tokens
    gen_tokens: List[int] = []
    gen_ctx: List[int] = []
    gen_tables: List[Sequence[int]] = []
    for g in seq_group_metadata_list:
        if g.is_prompt:
            continue
        seq_ids = list(g.seq_data.keys())
        seq_groups.append((seq_ids, g.sampling_params))
        for seq_id in seq_ids:
            data = g.seq_data[seq_id]
            gen_tokens.append(int(data.get_last_token_id()))
            gen_ctx.append(int(data.get_len()))
            gen_tables.append(g.block_tables[seq_id])
    num_gen = len(gen_ctx)
    ctx = np.asarray(gen_ctx, dtype=np.int64)
    max_context_len = int(ctx.max()) if num_gen else 0
    max_blocks = max((len(t) for t in gen_tables), default=0)
    tables = np.zeros((num_gen, max_blocks), dtype=np.int32)
    for i, t in enumerate(gen_tables):
        tables[i, :len(t)] = t
    if num_gen:
        pos = ctx - 1
        tok_parts.append(np.asarray(gen_tokens, dtype=np.int64))
        pos_parts.append(pos)
        rows = np.arange(num_gen)
        slot_parts.append(tables[rows, pos // B].astype(np.int64) * B + pos % B)

    tokens = np.concatenate(tok_parts) if tok_parts else np.zeros(0, dtype=np.int64)
    positions = np.concatenate(pos_parts) if pos_parts else np.zeros(0, dtype=np.int64)
    slots = (np.concatenate(slot_parts) if slot_parts else np.zeros(0, dtype=np.int64)).astype(np.int32)
    n_valid = int(tokens.shape[0])
    n_pad = _pad_len(n_valid, pad_multiple)

    # one staging buffer: [tokens i64 | positions i64 | slots i32 | ctx i32 | tables i32]
    off_tok = 0
    off_pos = _align(off_tok + 8 * n_pad)
    off_slot = _align(off_pos + 8 * n_pad)
    off_ctx = _align(off_slot + 4 * n_valid)
    off_tab = _align(off_ctx + 4 * num_gen)
    total = _align(off_tab + 4 * num_gen * max_blocks)
    device = torch.device(device)
    pinned = device.type == "cuda"
    host = torch.zeros(max(total, 16), dtype=torch.uint8, pin_memory=pinned)
    hb = host.numpy()
    hb[off_tok:off_tok + 8 * n_valid].view(np.int64)[:] = tokens
    hb[off_pos:off_pos + 8 * n_valid].view(np.int64)[:] = positions
    hb[off_slot:off_slot + 4 * n_valid].view(np.int32)[:] = slots
    hb[off_ctx:off_ctx + 4 * num_gen].view(np.int32)[:] = ctx.astype(np.int32)
    hb[off_tab:off_tab + 4 * num_gen * max_blocks].view(np.int32)[:] = tables.reshape(-1)
    dev = host.to(device, non_blocking=True) if pinned else host

    def view(off: int, nbytes: int, dtype: torch.dtype) -> torch.Tensor:
        return dev[off:off + nbytes].view(dtype)

    tokens_t = view(off_tok, 8 * n_pad, torch.int64)
    positions_t = view(off_pos, 8 * n_pad, torch.int64)
    slot_t = view(off_slot, 4 * n_valid, torch.int32)
    ctx_t = view(off_ctx, 4 * num_gen, torch.int32)
    tables_t = view(off_tab, 4 * num_gen * max_blocks, torch.int32).reshape(num_gen, max_blocks)

    seq_data: Dict[int, Any] = {}
    for g in seq_group_metadata_list:
        seq_data.update(g.seq_data)
    meta = InputMetadata(
        seq_groups=seq_groups,
        seq_data=seq_data,
        prompt_lens=prompt_lens,
        slot_mapping=slot_t,
        context_lens=ctx_t,
        max_context_len=max_context_len,
        block_tables=tables_t,
    )
    # keep the staging buffer alive until the copy has been consumed
    meta._staging = host
    # known here on the host, and useful to the attention operator's launch plan
    meta.total_context_len = int(ctx.sum()) if num_gen else 0
    # every generation token's slot is the last position of its context in its own block table
    # (computed so above): the attention layer may derive it there instead of reading slot_mapping
    meta.generation_slots_follow_tables = True
    return tokens_t, positions_t, meta
