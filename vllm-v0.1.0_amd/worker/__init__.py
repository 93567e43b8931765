from .cache_engine import KV_STAGGER_BYTES, CacheEngine, carve_kv_cache, kv_slab_elements, simple_configs  # noqa: F401
from .inputs import prepare_inputs  # noqa: F401
from .all_reduce import GraphAllReduce, Pending, get_all_reduce_launcher, initialize_all_reduce_launcher  # noqa: F401
from .tp_decode import TensorParallelDecodeStep, head_shard  # noqa: F401
from .push_all_reduce import PushAllReduce  # noqa: F401
