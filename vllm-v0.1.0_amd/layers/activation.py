"""SwiGLU activation over the HIP silu_and_mul operator.

Counterpart of SiluAndMul (vllm/model_executor/layers/activation.py:23-40):
x [num_tokens, 2 d] -> silu(x[:, :d]) * x[:, d:].
"""
import torch
import torch.nn as nn


def get_act_fn(act_fn: str) -> nn.Module:
    """Activation module by its model-config name (get_act_fn,
    vllm/model_executor/layers/activation.py:8-21): "gelu", "gelu_new" and "gelu_fast" (both the
    tanh approximation), "relu"; case-insensitive; anything else raises ValueError.  These run as
    torch's own element-wise kernels, as in the reference; only SiluAndMul has a HIP operator."""
    makers = {"gelu": nn.GELU, "gelu_new": lambda: nn.GELU(approximate="tanh"),
              "gelu_fast": lambda: nn.GELU(approximate="tanh"), "relu": nn.ReLU}
    try:
        return makers[act_fn.lower()]()
    except KeyError:
        raise ValueError(f"Activation function {act_fn!r} is not supported.") from None


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
