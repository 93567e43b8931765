"""The slab layout of the KV caches (worker/cache_engine.py): every layer's key and value cache are views of one
slab, none overlap, the shapes are the reference's (vllm/worker/cache_engine.py:48-63), and a layer's value
cache starts KV_STAGGER_BYTES behind the end of its key cache."""
import pytest


@pytest.mark.parametrize("dtype_name,x", [("float16", 8), ("bfloat16", 8), ("float32", 4)])
def test_layers_are_disjoint_views_with_the_stagger(pkg, dtype_name, x):
    import torch
    from vllm_amd.worker import KV_STAGGER_BYTES, carve_kv_cache, kv_slab_elements
    dtype = getattr(torch, dtype_name)
    es = torch.tensor([], dtype=dtype).element_size()
    L, NB, H, D, B = 3, 5, 2, 64, 16
    kshape, vshape = (H, D // x, B, x), (H, D, B)
    slab = torch.zeros(kv_slab_elements(L, NB, vshape, es), dtype=dtype)
    seen = set()
    for layer in range(L):
        k, v = carve_kv_cache(slab, layer, NB, kshape, vshape)
        assert k.shape == (NB, *kshape) and v.shape == (NB, *vshape) and k.is_contiguous() and v.is_contiguous()
        assert k.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0
        assert v.data_ptr() - (k.data_ptr() + k.numel() * es) == KV_STAGGER_BYTES
        for t, tag in ((k, 2 * layer + 1), (v, 2 * layer + 2)):
            lo = (t.data_ptr() - slab.data_ptr()) // es
            span = set(range(lo, lo + t.numel(), 997)) | {lo, lo + t.numel() - 1}
            assert not (span & seen)
            seen |= span
            t.fill_(tag)
    # every cache still holds its own tag: nothing was written through another view
    for layer in range(L):
        k, v = carve_kv_cache(slab, layer, NB, kshape, vshape)
        assert bool((k == 2 * layer + 1).all()) and bool((v == 2 * layer + 2).all())
    assert KV_STAGGER_BYTES % 16 == 0 and KV_STAGGER_BYTES % (32 << 10) != 0
