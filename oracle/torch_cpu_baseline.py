"""Torch-CPU statements of decode attention, for the cpu_baseline leg of bench.py.

TEST / MEASUREMENT INFRASTRUCTURE: only tests/ and bench.py's cpu_baseline leg
may import this module; the product path never does.

The reference has no CPU attention path; its only CPU-runnable statement of
the decode path is the kernel test's oracle
(/root/reference/tests/kernels/test_attention.py:14-66): per sequence, gather
the context's keys and values token by token out of the paged caches, then one
masked attention in torch.  Two restatements of it, both accumulating in fp32:

  decode_loop        the same shape of work: a Python loop over sequences and
                     over context tokens, one cache slice per token, then
                     einsum / softmax / einsum per sequence (:42-66, :14-27);
  decode_vectorised  what a torch user would write instead: one index_select
                     of whole blocks per sequence, a reshape into [ctx, H, D],
                     and the same two einsums.

Inputs are torch CPU tensors in the operator's layouts; out is [S, H, D] fp32.
"""
from __future__ import annotations

import torch


def _attend(q: torch.Tensor, keys: torch.Tensor, values: torch.Tensor, scale: float) -> torch.Tensor:
    """q [H, D], keys / values [ctx, H, D] -> [H, D]; fp32."""
    attn = torch.einsum("hd,khd->hk", q * scale, keys)
    attn = torch.softmax(attn, dim=-1)
    return torch.einsum("hk,khd->hd", attn, values)


def decode_loop(query, key_cache, value_cache, block_tables, context_lens, scale: float) -> torch.Tensor:
    S, H, D = query.shape
    B = value_cache.shape[3]
    out = torch.zeros(S, H, D, dtype=torch.float32)
    tables = block_tables.tolist()
    lens = context_lens.tolist()
    for s in range(S):
        keys, values = [], []
        for j in range(lens[s]):
            block = tables[s][j // B]
            off = j % B
            keys.append(key_cache[block, :, :, off, :].reshape(H, D))
            values.append(value_cache[block, :, :, off])
        if not keys:
            continue
        k = torch.stack(keys, dim=0).float()
        v = torch.stack(values, dim=0).float()
        out[s] = _attend(query[s].float(), k, v, scale)
    return out


def decode_vectorised(query, key_cache, value_cache, block_tables, context_lens, scale: float) -> torch.Tensor:
    S, H, D = query.shape
    B = value_cache.shape[3]
    out = torch.zeros(S, H, D, dtype=torch.float32)
    lens = context_lens.tolist()
    for s in range(S):
        n = lens[s]
        if n == 0:
            continue
        nb = (n + B - 1) // B
        idx = block_tables[s, :nb].long()
        # [nb, H, D/x, B, x] -> [nb, B, H, D/x, x] -> [nb * B, H, D]
        k = key_cache.index_select(0, idx).permute(0, 3, 1, 2, 4).reshape(nb * B, H, D)[:n].float()
        # [nb, H, D, B] -> [nb, B, H, D]
        v = value_cache.index_select(0, idx).permute(0, 3, 1, 2).reshape(nb * B, H, D)[:n].float()
        out[s] = _attend(query[s].float(), k, v, scale)
    return out
