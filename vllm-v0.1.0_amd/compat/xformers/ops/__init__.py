"""`xformers.ops` names used by vllm/model_executor/layers/attention.py:44-67."""
from __future__ import annotations

import importlib
from typing import Optional

import torch

from . import fmha  # noqa: F401
from .fmha.attn_bias import AttentionBias, BlockDiagonalCausalMask

__all__ = ["AttentionBias", "fmha", "memory_efficient_attention_forward"]


def _attention_ops():
    """The HIP operator module: inside a vLLM checkout it is `vllm.attention_ops`,
    in this repository the package registered as `vllm_amd`."""
    for name in ("vllm.attention_ops", "vllm_amd.attention_ops"):
        try:
            return importlib.import_module(name)
        except ImportError:
            continue
    raise ImportError("neither vllm.attention_ops nor vllm_amd.attention_ops is importable: "
                      "build the MI355X operator modules first")


def memory_efficient_attention_forward(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                                       attn_bias: Optional[AttentionBias] = None, p: float = 0.0,
                                       scale: Optional[float] = None, *, op=None) -> torch.Tensor:
    """query / key / value: [1, num_tokens, num_heads, head_size] (views of a
    fused qkv tensor are fine); attn_bias: BlockDiagonalCausalMask over the
    token axis.  Returns [1, num_tokens, num_heads, head_size]; rows the mask
    does not cover are zero."""
    if not isinstance(attn_bias, BlockDiagonalCausalMask):
        raise NotImplementedError("only BlockDiagonalCausalMask attention is available")
    if p != 0.0:
        raise NotImplementedError("dropout is not available")
    if query.dim() != 4 or query.shape[0] != 1:
        raise ValueError("expected [1, num_tokens, num_heads, head_size] tensors")
    if scale is None:
        scale = query.shape[-1] ** -0.5
    q, k, v = query[0], key[0], value[0]
    out = torch.zeros(q.shape, dtype=q.dtype, device=q.device)
    _attention_ops().multi_query_kv_attention(out, q, k, v, list(attn_bias.seqlens), float(scale))
    return out.unsqueeze(0)
