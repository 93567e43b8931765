"""SwiGLU activation over the HIP silu_and_mul operator.

Counterpart of SiluAndMul (vllm/model_executor/layers/activation.py:23-40):
x [num_tokens, 2 d] -> silu(x[:, :d]) * x[:, d:].
"""
import torch
import torch.nn as nn


class SiluAndMul(nn.Module):

    def __init__(self, activation_ops=None) -> None:
        super().__init__()
        if activation_ops is None:
            from .. import activation_ops  # HIP module; no fallback
        self._activation_ops = activation_ops

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty(x.shape[0], x.shape[1] // 2, dtype=x.dtype, device=x.device)
        self._activation_ops.silu_and_mul(out, x)
        return out
