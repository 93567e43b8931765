from .activation import SiluAndMul, get_act_fn  # noqa: F401
from .attention import PagedAttention, PagedAttentionWithRoPE  # noqa: F401
from .layernorm import RMSNorm  # noqa: F401
