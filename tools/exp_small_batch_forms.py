#!/usr/bin/env python3
"""Experiment: launch forms of small decode batches through the per-call entry point
(pa_single_query_cached_kv_attention_ws: what the module function with the reference's signature calls).

    PA_DECODE_SPLIT_TOKENS=... PA_DECODE_TARGET_WGS=... tools/exp_small_batch_forms.py NAME

NAME: a `make variant` build (reads the PA_DECODE_* knobs).  Times graph replays of 10 launches over
rotating cache copies; 32 heads x 128, block 16, fp16."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

name = sys.argv[1]
path = os.path.join(ROOT, "vllm-v0.1.0_amd", "libpaged_attn_mi355x.so" if name == "default" else f"tune/libpa_{name}.so")
lib = ctypes.CDLL(path)
vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
lib.pa_single_query_cached_kv_attention_ws.argtypes = [vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, vp,
                                                       ctypes.c_size_t, vp]
lib.pa_decode_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
lib.pa_decode_workspace_bytes.restype = ctypes.c_size_t
dev = torch.device("cuda:0")
H, D, B = 32, 128, 16
shapes = [(1, 1024), (2, 1024), (3, 1024), (4, 1024), (5, 1024), (6, 1024), (7, 1024), (8, 1024), (12, 1024), (16, 1024),
          (4, 512), (4, 768), (6, 512), (8, 512), (8, 2048), (1, 128), (8, 128)]
knobs = {k: v for k, v in os.environ.items() if k.startswith("PA_DECODE_")}
print(f"# {name} {knobs}")
for S, ctx in shapes:
    bps = ctx // B
    NB = S * bps
    copies = 8
    kcs = [torch.empty(NB, H, D // 8, B, 8, dtype=torch.float16, device=dev).normal_(0, .5) for _ in range(copies)]
    vcs = [torch.empty(NB, H, D, B, dtype=torch.float16, device=dev).normal_(0, .5) for _ in range(copies)]
    g = torch.Generator(); g.manual_seed(0)
    bt = torch.randperm(NB, generator=g).to(torch.int32).reshape(S, bps).to(dev)
    cl = torch.full((S,), ctx, dtype=torch.int32, device=dev)
    qkv = torch.empty(S, 3, H, D, dtype=torch.float16, device=dev).normal_(0, .5)
    q = qkv[:, 0]
    out = torch.empty(S, H, D, dtype=torch.float16, device=dev)
    nb = lib.pa_decode_workspace_bytes(S, H, D, B, ctx)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)

    def call(i, stream):
        rc = lib.pa_single_query_cached_kv_attention_ws(
            out.data_ptr(), q.data_ptr(), kcs[i % copies].data_ptr(), vcs[i % copies].data_ptr(), D ** -0.5, bt.data_ptr(),
            cl.data_ptr(), S, H, D, B, bps, ctx, q.stride(0), 1, ws.data_ptr(), nb, ctypes.c_void_p(stream.cuda_stream))
        assert rc == 0

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(10):
            call(i, side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for i in range(10):
            call(i, side)
    graph.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graph.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 10 * 1e3)
    ts.sort()
    mb = 2 * S * H * ctx * D * 2 / 1e6
    print(f"bs {S:3d} ctx {ctx:5d}: {ts[len(ts) // 2]:7.2f} us (min {ts[0]:7.2f})  {100 * mb / ts[len(ts) // 2] / 8e6:5.1f} % of 8 TB/s", flush=True)
    del kcs, vcs
