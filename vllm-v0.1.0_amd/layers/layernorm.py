"""RMSNorm layer over the HIP rms_norm operator.

Counterpart of vllm/model_executor/layers/layernorm.py:8-32: x -> w * x /
sqrt(mean(x^2) + eps), statistics in fp32, result in the input's dtype.
"""
import torch
import torch.nn as nn


class RMSNorm(nn.Module):

    def __init__(self, hidden_size: int, eps: float = 1e-6, layernorm_ops=None) -> None:
        super().__init__()
        if layernorm_ops is None:
            from .. import layernorm_ops  # HIP module; no fallback
        self._layernorm_ops = layernorm_ops
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(x)
        self._layernorm_ops.rms_norm(out, x, self.weight.data, self.variance_epsilon)
        return out
