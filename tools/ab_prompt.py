#!/usr/bin/env python3
"""In-process A/B of prompt-attention builds: loads the product library and
`make variant` builds (vllm-v0.1.0_amd/tune/libpa_<name>.so) and times them in
interleaved rounds on the same inputs (graph replay of 3 launches).

    python tools/ab_prompt.py --libs default,x [--dtype bfloat16] [--shapes 1x16384,8x2048,1x2560,5x512]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

PKG = os.path.join(ROOT, "vllm-v0.1.0_amd")


def load(name):
    path = os.path.join(PKG, "libpaged_attn_mi355x.so") if name == "default" else os.path.join(PKG, "tune", f"libpa_{name}.so")
    lib = ctypes.CDLL(path)
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    lib.pa_multi_query_kv_attention.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f32, i64, i64, i64, i32, vp]
    lib.pa_multi_query_kv_attention.restype = i32
    lib.pa_last_error.restype = ctypes.c_char_p
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="default")
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--shapes", default="1x16384,4x4096,8x2048,1x2560,5x512")
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--head-size", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    names = args.libs.split(",")
    libs = {n: load(n) for n in names}
    dt = getattr(torch, args.dtype)
    code = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dt]
    H, D = args.heads, args.head_size
    torch.manual_seed(0)
    for shape in args.shapes.split(","):
        ns, n = (int(x) for x in shape.split("x"))
        T = ns * n
        qkv = torch.randn(T, 3, H, D, dtype=dt, device="cuda:0")
        out = torch.empty(T, H, D, dtype=dt, device="cuda:0")
        cu = (ctypes.c_int32 * (ns + 1))(*[i * n for i in range(ns + 1)])
        flops = 4 * H * D * ns * n * (n + 1) // 2
        graphs, ref = {}, None
        for name, lib in libs.items():
            def call():
                st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                rc = lib.pa_multi_query_kv_attention(out.data_ptr(), qkv[:, 0].data_ptr(), qkv[:, 1].data_ptr(), qkv[:, 2].data_ptr(),
                                                     cu, ns, H, D, D ** -0.5, 3 * H * D, 3 * H * D, 3 * H * D, code, st)
                assert rc == 0, lib.pa_last_error()
            call(); torch.cuda.synchronize()
            chk = out.float().abs().mean().item()
            if ref is None:
                ref = out.clone()
                same = "reference"
            else:
                same = "identical" if torch.equal(out, ref) else f"max diff {(out.float() - ref.float()).abs().max().item():.3e}"
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                call()
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(3): call()
            graphs[name] = (g, chk, same)
        times = {n_: [] for n_ in names}
        for _ in range(args.rounds):
            for name in names:
                g = graphs[name][0]
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); g.replay(); b.record(); torch.cuda.synchronize()
                times[name].append(a.elapsed_time(b) / 3 * 1e3)
        for name in names:
            t = sorted(times[name])[len(times[name]) // 2]
            print(f"{shape:9s} {args.dtype:9s} {name:12s} {t:9.1f} us  {flops / t / 1e6:7.1f} TFLOP/s   mean|out| {graphs[name][1]:.5f}  {graphs[name][2]}", flush=True)

main()
