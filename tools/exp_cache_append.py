#!/usr/bin/env python3
"""Experiment: launch shape of the whole-tile value update of reshape_and_cache (decode-sized batches).

    tools/exp_cache_append.py NAME        NAME: a `make variant` build (reads PA_CACHE_APPEND_IN_FLIGHT /
                                          PA_CACHE_APPEND_WORKGROUPS from the environment)

Prints, for batch 256 / 64 / 16 at LLaMA-7B sizes: reshape_and_cache per launch (graph replay over rotating
layers), and the decode launch that follows it (HIP events), against the decode launch alone."""
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
dev = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else None
lib = ctypes.CDLL(os.path.join(ROOT, "vllm-v0.1.0_amd", "tune", f"libpa_{name}.so")) if name else pkg._capi.lib
vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
lib.pa_single_query_cached_kv_attention.argtypes = [vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, vp]
lib.pa_reshape_and_cache.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, vp]
H, D, B, CTX, L = 32, 128, 16, 1024, 6
bps = CTX // B


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(S):
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

    def dec(l):
        assert lib.pa_single_query_cached_kv_attention(out.data_ptr(), q.data_ptr(), kcs[l].data_ptr(), vcs[l].data_ptr(),
                                                       D ** -0.5, bt.data_ptr(), cl.data_ptr(), S, H, D, B, bps, CTX,
                                                       q.stride(0), 1, stream()) == 0

    def rsh(l):
        assert lib.pa_reshape_and_cache(k.data_ptr(), v.data_ptr(), kcs[l].data_ptr(), vcs[l].data_ptr(), slots.data_ptr(),
                                        S, H, D, B, k.stride(0), v.stride(0), 1, stream()) == 0

    # correctness of the variant against the shipped library's operator
    ref_k, ref_v = kcs[0].clone(), vcs[0].clone()
    pkg.cache_ops.reshape_and_cache(k, v, ref_k, ref_v, slots)
    rsh(0)
    torch.cuda.synchronize()
    assert torch.equal(ref_k, kcs[0]) and torch.equal(ref_v, vcs[0]), "variant differs from the shipped operator"

    for _ in range(3):
        dec(0)
    torch.cuda.synchronize()

    def decode_after(pre):
        ts = []
        for i in range(24):
            l = i % (L - 1)
            pre(l)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); dec(l); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        return statistics.median(ts[4:])

    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(10):
            rsh(i % L)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(10):
            rsh(i % L)
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 10 * 1e3)
    # the pair inside a step, eager: reshape + decode back to back over the layers
    pair = []
    for rep in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for l in range(L):
            rsh(l); dec(l)
        b.record(); torch.cuda.synchronize()
        pair.append(a.elapsed_time(b) / L * 1e3)
    knobs = {k_: os.environ[k_] for k_ in ("PA_CACHE_APPEND_IN_FLIGHT", "PA_CACHE_APPEND_WORKGROUPS") if k_ in os.environ}
    print(f"S={S:4d} {knobs}: reshape {statistics.median(ts):6.1f} us | decode alone {decode_after(lambda l: None):7.1f} "
          f"after reshape {decode_after(rsh):7.1f} | pair per layer {statistics.median(pair[1:]):7.1f} us", flush=True)


for S in (256, 64, 16):
    run(S)
