#!/usr/bin/env python3
"""Experiment: cost of reshape_and_cache on the decode kernel that follows."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, statistics
import __graft_entry__
pkg = __graft_entry__.load_package()
dev = "cuda:0"
S, H, D, B, CTX, L = int(os.environ.get("EXP_S", 256)), 32, 128, 16, int(os.environ.get("EXP_CTX", 1024)), 6
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
if len(sys.argv) > 1:  # decode kernel of a `make variant` build: tools/exp_step.py NAME
    import ctypes
    vlib = ctypes.CDLL(os.path.join(ROOT, "vllm-v0.1.0_amd", "tune", f"libpa_{sys.argv[1]}.so"))
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    vlib.pa_single_query_cached_kv_attention.argtypes = [vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, vp]
    def dec(l):
        rc = vlib.pa_single_query_cached_kv_attention(out.data_ptr(), q.data_ptr(), kcs[l].data_ptr(), vcs[l].data_ptr(), scale,
                                                      bt.data_ptr(), cl.data_ptr(), S, H, D, B, bps, CTX, q.stride(0), 1,
                                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    vlib.pa_reshape_and_cache.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, vp]
    def rsh(l):
        rc = vlib.pa_reshape_and_cache(k.data_ptr(), v.data_ptr(), kcs[l].data_ptr(), vcs[l].data_ptr(), slots.data_ptr(),
                                       S, H, D, B, k.stride(0), v.stride(0), 1,
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
else:
    rsh = lambda l: pkg.cache_ops.reshape_and_cache(k, v, kcs[l], vcs[l], slots)
    dec = lambda l: pkg.attention_ops.single_query_cached_kv_attention(out, q, kcs[l], vcs[l], scale, bt, cl, B, CTX)

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


def time_reshape():
    ts = []
    for i in range(24):
        l = i % (L - 1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4): rsh(l)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / 4)
    print(f"{'reshape_and_cache alone (4 back to back)':50s} median {statistics.median(ts[4:]):7.1f} us")
time_reshape()

def step_time():
    def steps(n):
        for _ in range(n):
            for l in range(L - 1):
                rsh(l); dec(l)
    steps(1); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); steps(6); b.record(); torch.cuda.synchronize()
    print(f"{'reshape + decode per layer, back to back':50s} {a.elapsed_time(b) * 1e3 / (6 * (L - 1)):7.1f} us")
step_time()
