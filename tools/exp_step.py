#!/usr/bin/env python3
"""Experiment: cost of reshape_and_cache on the decode kernel that follows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, statistics
import __graft_entry__
pkg = __graft_entry__.load_package()
dev = "cuda:0"
S, H, D, B, CTX, L = 256, 32, 128, 16, 1024, 6
bps = CTX // B
NB = S * bps
kcs = [torch.empty(NB, H, D // 8, B, 8, dtype=torch.float16, device=dev).normal_(0, .5) for _ in range(L)]
vcs = [torch.empty(NB, H, D, B, dtype=torch.float16, device=dev).normal_(0, .5) for _ in range(L)]
qkv = torch.empty(S, 3, H, D, dtype=torch.float16, device=dev).normal_(0, .5)
q, k, v = qkv.unbind(1)
out = torch.empty(S, H, D, dtype=torch.float16, device=dev)
g = torch.Generator(); g.manual_seed(0)
bt = torch.randperm(NB, generator=g).to(torch.int32).reshape(S, bps).to(dev)
cl = torch.full((S,), CTX, dtype=torch.int32, device=dev)
slots = (bt[:, -1].long() * B + 15).to(torch.int32)
last = bt[:, -1].long()
scale = D ** -0.5
dec = lambda l: pkg.attention_ops.single_query_cached_kv_attention(out, q, kcs[l], vcs[l], scale, bt, cl, B, CTX)
rsh = lambda l: pkg.cache_ops.reshape_and_cache(k, v, kcs[l], vcs[l], slots)

def measure(name, pre):
    ts = []
    for i in range(24):
        l = i % (L - 1)
        pre(l)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); dec(l); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print(f"{name:50s} decode median {statistics.median(ts[4:]):7.1f} us")

for _ in range(3): dec(0)
torch.cuda.synchronize()
measure("A decode alone", lambda l: None)
measure("B reshape(same layer) then decode", lambda l: rsh(l))

