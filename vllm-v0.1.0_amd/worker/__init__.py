from .cache_engine import CacheEngine, simple_configs  # noqa: F401
from .inputs import prepare_inputs  # noqa: F401
