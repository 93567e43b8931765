"""KV-cache owner: allocates the paged GPU and pinned-CPU caches and issues
swap / copy operations.

Counterpart of vllm/worker/cache_engine.py:12-150: same class, constructor
arguments, attributes (`gpu_cache`, `cpu_cache`, `cache_stream`, `events`) and
methods (`get_key_block_shape`, `get_value_block_shape`, `swap_in`, `swap_out`,
`copy`, `get_cache_block_size`), so vllm/worker/worker.py:120-126, 247-262
drives it unchanged.  The config arguments are duck-typed: anything with the
reference's accessors works (`CacheConfig.block_size / num_gpu_blocks /
num_cpu_blocks`, `ModelConfig.dtype / get_head_size() / get_num_layers(pc) /
get_num_heads(pc)`); `simple_configs()` builds minimal stand-ins.

Sized for a 288 GB part: every layer's K and V blocks live in ONE device slab
(and one pinned host slab), carved into the per-layer tensors the callers
see.  One allocation instead of 2 x num_layers keeps the allocator out of the
way when the cache takes most of HBM, and gives every layer the same
alignment — and the same distance between its key and its value cache, which
matters: the decode kernel fetches the key tile and the value tile of one
(block, head) together, from the same offset in the two caches, and when the
caches lie a multiple of 32 KiB apart (as two block-sized allocations do) the
two tiles fall on the same memory channels.  Measured on MI355X, 256 x 32 x
128, context 1024: 637.7 us per decode launch with the value cache right after
the key cache, 630.6 us with it KV_STAGGER_BYTES further on (4 to 8 KiB are
equally good, 32 KiB, 64 KiB and 1 MiB as bad as none).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Tuple

import torch

KVCache = Tuple[torch.Tensor, torch.Tensor]

# gap between the end of a layer's key cache and the start of its value cache (see above)
KV_STAGGER_BYTES = 8192


def carve_kv_cache(slab: torch.Tensor, layer: int, num_blocks: int, key_block_shape, value_block_shape) -> KVCache:
    """Layer `layer`'s (key cache, value cache) views of a flat slab holding kv_slab_elements(...) elements."""
    per_tensor = num_blocks
    for d in value_block_shape:
        per_tensor *= d
    stagger = KV_STAGGER_BYTES // slab.element_size()
    base = layer * (2 * per_tensor + stagger)
    k = slab[base:base + per_tensor].view(num_blocks, *key_block_shape)
    v = slab[base + per_tensor + stagger:base + 2 * per_tensor + stagger].view(num_blocks, *value_block_shape)
    return k, v


def kv_slab_elements(num_layers: int, num_blocks: int, value_block_shape, element_size: int) -> int:
    per_tensor = num_blocks
    for d in value_block_shape:
        per_tensor *= d
    return num_layers * (2 * per_tensor + KV_STAGGER_BYTES // element_size)


def _dtype_size(dtype: torch.dtype) -> int:
    return torch.tensor([], dtype=dtype).element_size()


def simple_configs(num_layers: int, num_heads: int, head_size: int, dtype: torch.dtype, block_size: int,
                   num_gpu_blocks: int, num_cpu_blocks: int):
    """Minimal (cache_config, model_config, parallel_config) triple with the
    accessors CacheEngine uses (vllm/config.py:70-85, 101-118)."""
    cache_config = SimpleNamespace(block_size=block_size, num_gpu_blocks=num_gpu_blocks,
                                   num_cpu_blocks=num_cpu_blocks)
    model_config = SimpleNamespace(
        dtype=dtype,
        get_head_size=lambda: head_size,
        get_num_layers=lambda parallel_config=None: num_layers,
        get_num_heads=lambda parallel_config=None: num_heads,
    )
    parallel_config = SimpleNamespace(tensor_parallel_size=1, pipeline_parallel_size=1, world_size=1)
    return cache_config, model_config, parallel_config


class CacheEngine:

    def __init__(self, cache_config, model_config, parallel_config, cache_ops=None, device="cuda") -> None:
        if cache_ops is None:
            from .. import cache_ops as _ops  # the HIP module; no fallback
            cache_ops = _ops
        self._ops = cache_ops
        self.cache_config = cache_config
        self.model_config = model_config
        self.parallel_config = parallel_config
        self.device = torch.device(device)

        self.head_size = model_config.get_head_size()
        self.num_layers = model_config.get_num_layers(parallel_config)
        self.num_heads = model_config.get_num_heads(parallel_config)
        self.dtype = model_config.dtype

        self.block_size = cache_config.block_size
        self.num_gpu_blocks = cache_config.num_gpu_blocks
        self.num_cpu_blocks = cache_config.num_cpu_blocks

        self.gpu_cache = self.allocate_gpu_cache()
        self.cpu_cache = self.allocate_cpu_cache()

        # swaps run on a side stream; one event per layer tells the attention
        # layer when that layer's blocks have arrived (attention.py:125-126)
        self.cache_stream = torch.cuda.Stream(device=self.device)
        assert self.cache_stream != torch.cuda.current_stream(self.device)
        self.events = [torch.cuda.Event() for _ in range(self.num_layers)]

    def get_key_block_shape(self) -> Tuple[int, int, int, int]:
        x = 16 // _dtype_size(self.dtype)
        return (self.num_heads, self.head_size // x, self.block_size, x)

    def get_value_block_shape(self) -> Tuple[int, int, int]:
        return (self.num_heads, self.head_size, self.block_size)

    def _allocate(self, num_blocks: int, **kw) -> List[KVCache]:
        kshape, vshape = self.get_key_block_shape(), self.get_value_block_shape()
        slab = torch.empty(kv_slab_elements(self.num_layers, num_blocks, vshape, _dtype_size(self.dtype)),
                           dtype=self.dtype, **kw)
        return [carve_kv_cache(slab, layer, num_blocks, kshape, vshape) for layer in range(self.num_layers)]

    def allocate_gpu_cache(self) -> List[KVCache]:
        return self._allocate(self.num_gpu_blocks, device=self.device)

    def allocate_cpu_cache(self) -> List[KVCache]:
        return self._allocate(self.num_cpu_blocks, pin_memory=True)

    def _swap(self, src: List[KVCache], dst: List[KVCache], src_to_dst: Dict[int, int]) -> None:
        with torch.cuda.stream(self.cache_stream):
            for i in range(self.num_layers):
                src_key_cache, src_value_cache = src[i]
                dst_key_cache, dst_value_cache = dst[i]
                self._ops.swap_blocks(src_key_cache, dst_key_cache, src_to_dst)
                self._ops.swap_blocks(src_value_cache, dst_value_cache, src_to_dst)
                self.events[i].record(stream=self.cache_stream)

    def swap_in(self, src_to_dst: Dict[int, int]) -> None:
        self._swap(self.cpu_cache, self.gpu_cache, src_to_dst)

    def swap_out(self, src_to_dst: Dict[int, int]) -> None:
        self._swap(self.gpu_cache, self.cpu_cache, src_to_dst)

    def copy(self, src_to_dsts: Dict[int, List[int]]) -> None:
        key_caches = [key_cache for key_cache, _ in self.gpu_cache]
        value_caches = [value_cache for _, value_cache in self.gpu_cache]
        # asynchronous on the current stream (the reference's copy_blocks
        # blocks the host, cache_kernels.cu:116-121)
        self._ops.copy_blocks(key_caches, value_caches, src_to_dsts)

    @staticmethod
    def get_cache_block_size(block_size: int, model_config, parallel_config) -> int:
        head_size = model_config.get_head_size()
        num_heads = model_config.get_num_heads(parallel_config)
        num_layers = model_config.get_num_layers(parallel_config)
        key_cache_block = block_size * num_heads * head_size
        value_cache_block = key_cache_block
        total = num_layers * (key_cache_block + value_cache_block)
        return _dtype_size(model_config.dtype) * total
