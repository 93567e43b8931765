from . import attn_bias, cutlass  # noqa: F401
