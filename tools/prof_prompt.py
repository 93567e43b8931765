#!/usr/bin/env python3
"""One prompt-attention shape, a few plain launches: the target of rocprofv3
counter passes.  usage: prof_prompt.py [dtype] [n_prompts] [len] [heads] [head_size]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__
pkg = __graft_entry__.load_package()
dt = getattr(torch, sys.argv[1]) if len(sys.argv) > 1 else torch.bfloat16
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
H = int(sys.argv[4]) if len(sys.argv) > 4 else 32
D = int(sys.argv[5]) if len(sys.argv) > 5 else 128
T = ns * n
torch.manual_seed(0)
qkv = torch.randn(T, 3, H, D, dtype=dt, device="cuda:0")
q, k, v = qkv.unbind(dim=1)
out = torch.empty(T, H, D, dtype=dt, device="cuda:0")
for _ in range(5):
    pkg.attention_ops.multi_query_kv_attention(out, q, k, v, [n] * ns, D ** -0.5)
torch.cuda.synchronize()
print("done", float(out.float().abs().mean()))
