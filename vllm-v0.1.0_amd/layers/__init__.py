from .attention import PagedAttention  # noqa: F401
