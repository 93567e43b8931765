"""Caches beyond 4 GiB: every operator must address blocks with 64-bit offsets.

One K and one V cache of 5 GiB each (LLaMA-7B block geometry, 40 960 blocks of
128 KiB), with all traffic aimed at blocks past the 4 GiB mark.  The expected
results come from the same operators run on a compact copy of just those
blocks (block numbers remapped), which the other tests pin against the
oracle: equality here is bit for bit.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
H, D, B, X = 32, 128, 16, 8
BLOCK_BYTES = H * D * B * 2
NB = (5 << 30) // BLOCK_BYTES
FIRST_HIGH = (4 << 30) // BLOCK_BYTES + 1   # every byte of these blocks lies past 4 GiB


@pytest.fixture(scope="module")
def big(pkg):
    import torch
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < (12 << 30):
        pytest.skip("needs 12 GiB of free device memory")
    kc = torch.empty(NB, H, D // X, B, X, dtype=torch.float16, device=DEV)
    vc = torch.empty(NB, H, D, B, dtype=torch.float16, device=DEV)
    assert kc.numel() * 2 > (1 << 32) and vc.numel() * 2 > (1 << 32)
    yield kc, vc
    del kc, vc
    torch.cuda.empty_cache()


def _high_blocks(rng, n):
    return rng.permutation(np.arange(FIRST_HIGH, NB))[:n].astype(np.int64)


def test_decode_reads_blocks_past_4gib(pkg, big):
    import torch
    kc, vc = big
    rng = np.random.default_rng(5)
    ctx = [300, 17, 1024, 555, 1, 64, 2500]
    S = len(ctx)
    nblk = [(c + B - 1) // B for c in ctx]
    width = max(nblk) + 2
    ids = _high_blocks(rng, sum(nblk))
    ids_t = torch.from_numpy(ids).to(DEV)
    g = torch.Generator(device=DEV)
    g.manual_seed(6)
    kc[ids_t] = torch.randn(len(ids), H, D // X, B, X, dtype=torch.float16, device=DEV, generator=g) * 0.5
    vc[ids_t] = torch.randn(len(ids), H, D, B, dtype=torch.float16, device=DEV, generator=g) * 0.5
    kc_small, vc_small = kc[ids_t].clone(), vc[ids_t].clone()
    bt_big = np.zeros((S, width), dtype=np.int32)
    bt_small = np.zeros((S, width), dtype=np.int32)
    n0 = 0
    for s in range(S):
        bt_big[s, :nblk[s]] = ids[n0:n0 + nblk[s]]
        bt_small[s, :nblk[s]] = np.arange(n0, n0 + nblk[s])
        n0 += nblk[s]
    cl = torch.tensor(ctx, dtype=torch.int32, device=DEV)
    q = torch.randn(S, H, D, dtype=torch.float16, device=DEV, generator=g)
    scale = D ** -0.5
    for split in (False, True):
        outs = []
        for k_, v_, bt in ((kc, vc, bt_big), (kc_small, vc_small, bt_small)):
            out = torch.full((S, H, D), float("nan"), dtype=torch.float16, device=DEV)
            bt_t = torch.from_numpy(bt).to(DEV)
            if split:
                nbytes = pkg._capi.decode_workspace_bytes(S, H, D, B, max(ctx))
                assert nbytes > 0
                ws = torch.empty(nbytes // 4, dtype=torch.float32, device=DEV)
                pkg._capi.single_query_cached_kv_attention_ws(out, q, k_, v_, scale, bt_t, cl, B, max(ctx), ws)
            else:
                pkg._capi.single_query_cached_kv_attention(out, q, k_, v_, scale, bt_t, cl, B, max(ctx))
            torch.cuda.synchronize()
            outs.append(out)
        assert not bool(torch.isnan(outs[0]).any())
        assert torch.equal(outs[0], outs[1]), f"split={split}"


@pytest.mark.parametrize("T", [64, 1536], ids=["decode_sized", "prompt_sized"])
def test_reshape_gather_copy_past_4gib(pkg, big, T):
    import torch
    kc, vc = big
    rng = np.random.default_rng(7 + T)
    nb = (T + B - 1) // B + 4
    ids = _high_blocks(rng, 2 * nb)
    src_ids, dst_ids = ids[:nb], ids[nb:]
    ids_t = torch.from_numpy(ids).to(DEV)
    kc[ids_t] = 0
    vc[ids_t] = 0
    g = torch.Generator(device=DEV)
    g.manual_seed(8)
    qkv = torch.randn(T, 3, H, D, dtype=torch.float16, device=DEV, generator=g)
    # consecutive slots through the scattered blocks (a prompt), every block full but the last
    slots_small = np.arange(T, dtype=np.int64)
    slots_big = src_ids[slots_small // B] * B + slots_small % B
    assert slots_big.max() < 2 ** 31
    sb = torch.from_numpy(slots_big.astype(np.int32)).to(DEV)
    ss = torch.from_numpy(slots_small.astype(np.int32)).to(DEV)
    kc_small = torch.zeros(nb, H, D // X, B, X, dtype=torch.float16, device=DEV)
    vc_small = torch.zeros(nb, H, D, B, dtype=torch.float16, device=DEV)
    pkg.cache_ops.reshape_and_cache(qkv[:, 1], qkv[:, 2], kc, vc, sb)
    pkg.cache_ops.reshape_and_cache(qkv[:, 1], qkv[:, 2], kc_small, vc_small, ss)
    torch.cuda.synchronize()
    src_t = torch.from_numpy(src_ids).to(DEV)
    assert torch.equal(kc[src_t], kc_small) and torch.equal(vc[src_t], vc_small)
    back = torch.zeros_like(qkv)
    pkg.cache_ops.gather_cached_kv(back[:, 1], back[:, 2], kc, vc, sb)
    torch.cuda.synchronize()
    assert torch.equal(back[:, 1], qkv[:, 1]) and torch.equal(back[:, 2], qkv[:, 2])
    # copy-on-write between blocks that both lie past 4 GiB
    mapping = {int(s): [int(d)] for s, d in zip(src_ids, dst_ids)}
    pkg.cache_ops.copy_blocks([kc], [vc], mapping)
    torch.cuda.synchronize()
    dst_t = torch.from_numpy(dst_ids).to(DEV)
    assert torch.equal(kc[dst_t], kc_small) and torch.equal(vc[dst_t], vc_small)


def test_swap_blocks_past_4gib(pkg, big):
    import torch
    kc, _ = big
    rng = np.random.default_rng(9)
    ids = _high_blocks(rng, 6)
    ids_t = torch.from_numpy(ids).to(DEV)
    g = torch.Generator(device=DEV)
    g.manual_seed(10)
    kc[ids_t] = torch.randn(6, H, D // X, B, X, dtype=torch.float16, device=DEV, generator=g)
    want = kc[ids_t].cpu()
    cpu = torch.zeros(6, H, D // X, B, X, dtype=torch.float16).pin_memory()
    pkg.cache_ops.swap_blocks(kc, cpu, {int(b): i for i, b in enumerate(ids)})
    torch.cuda.synchronize()
    assert torch.equal(cpu, want)
    kc[ids_t] = 0
    pkg.cache_ops.swap_blocks(cpu, kc, {i: int(b) for i, b in enumerate(ids)})
    torch.cuda.synchronize()
    assert torch.equal(kc[ids_t].cpu(), want)
