import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import __graft_entry__
pkg = __graft_entry__.load_package()
dev = "cuda:0"

def weights(D, B, ctx, dtype):
    """Recovers the effective softmax weight of every token for K = 0."""
    x = 16 // torch.tensor([], dtype=dtype).element_size()
    nb = (ctx + B - 1) // B
    q = torch.ones(1, 1, D, dtype=dtype, device=dev)
    kc = torch.zeros(nb, 1, D // x, B, x, dtype=dtype, device=dev)
    bt = torch.arange(nb, dtype=torch.int32, device=dev)[None, :]
    cl = torch.tensor([ctx], dtype=torch.int32, device=dev)
    res = []
    for tt in range(nb * B):
        vc = torch.zeros(nb, 1, D, B, dtype=dtype, device=dev)
        vc[tt // B, 0, :, tt % B] = 1.0
        out = torch.full((1, 1, D), float("nan"), dtype=dtype, device=dev)
        pkg._capi.single_query_cached_kv_attention(out, q, kc, vc, 1.0, bt, cl, B, ctx)
        torch.cuda.synchronize()
        o = out.float().cpu().numpy()[0, 0]
        res.append((o.min() * ctx, o.max() * ctx))
    return res

for dtype in (torch.float16, torch.bfloat16):
    for (D, B, ctx) in ((128, 16, 22), (64, 8, 13), (128, 32, 40)):
        r = weights(D, B, ctx, dtype)
        print(dtype, D, B, ctx, " ".join(f"{a:.2f}/{b:.2f}" for a, b in r))
