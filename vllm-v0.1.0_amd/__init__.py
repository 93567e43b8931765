"""MI355X-native PagedAttention decode path (drop-in for vLLM v0.1.0's
`vllm.attention_ops` and `vllm.cache_ops`).

The directory name is not a Python identifier; load it with
`__graft_entry__.load_package()` (it registers the package as `vllm_amd`), or
copy the `*_ops*.so` modules and `libpaged_attn_mi355x.so` into a `vllm/`
package directory, where `from vllm import attention_ops, cache_ops,
pos_encoding_ops, layernorm_ops, activation_ops` then resolves to them.

Importing the package loads the HIP library and the operator modules and
fails if any of them has not been built: there is no CPU or eager fallback.
"""
import torch  # noqa: F401  (libtorch and the HIP runtime must be loaded first)

from . import _capi  # noqa: E402,F401
from . import attention_ops  # noqa: E402,F401
from . import cache_ops  # noqa: E402,F401
from . import pos_encoding_ops  # noqa: E402,F401
from . import layernorm_ops  # noqa: E402,F401
from . import activation_ops  # noqa: E402,F401

__all__ = ["attention_ops", "cache_ops", "pos_encoding_ops", "layernorm_ops", "activation_ops", "_capi"]
