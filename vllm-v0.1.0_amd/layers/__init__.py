from .activation import SiluAndMul  # noqa: F401
from .attention import PagedAttention, PagedAttentionWithRoPE  # noqa: F401
from .layernorm import RMSNorm  # noqa: F401
