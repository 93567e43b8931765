#!/usr/bin/env python3
"""In-process A/B of decode-kernel builds: loads several builds of the C-ABI
library (the product one plus `make variant` builds under
vllm-v0.1.0_amd/tune/), and times them in interleaved rounds on the same
inputs.  Prints median and min launch time per build and shape.

    python tools/ab_decode.py --libs default,nt [--shapes north_star,c2,ragged] [--sequential]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import algorithmic_bytes  # noqa: E402

PKG = os.path.join(ROOT, "vllm-v0.1.0_amd")

SHAPES = {
    "north_star": dict(S=256, H=32, D=128, B=16, ctx=1024, ragged=False),
    "c2": dict(S=256, H=12, D=64, B=16, ctx=1024, ragged=False),
    "ragged": dict(S=256, H=32, D=128, B=16, ctx=1024, ragged=True),
    "head96": dict(S=256, H=32, D=96, B=16, ctx=1024, ragged=False),
    "head80": dict(S=256, H=32, D=80, B=16, ctx=1024, ragged=False),
    "head64": dict(S=256, H=32, D=64, B=16, ctx=1024, ragged=False),
    "c2_ragged": dict(S=256, H=12, D=64, B=16, ctx=1024, ragged=True),
    "c2_block8": dict(S=256, H=12, D=64, B=8, ctx=1024, ragged=False),
    "c2_bs16": dict(S=16, H=12, D=64, B=16, ctx=1024, ragged=False),
    "bs16": dict(S=16, H=32, D=128, B=16, ctx=1024, ragged=False),
    "bs1": dict(S=1, H=32, D=128, B=16, ctx=1024, ragged=False),
    "block8": dict(S=256, H=32, D=128, B=8, ctx=1024, ragged=False),
    "block32": dict(S=256, H=32, D=128, B=32, ctx=1024, ragged=False),
    "ctx4k": dict(S=64, H=32, D=128, B=16, ctx=4096, ragged=False),
    # groups of 3 sequences share the physical blocks of the first half of the context
    "share3": dict(S=256, H=16, D=96, B=16, ctx=1024, ragged=False, share=3),
    "share3_ns": dict(S=255, H=32, D=128, B=16, ctx=1024, ragged=False, share=3),
    "ragged2048": dict(S=256, H=32, D=128, B=16, ctx=2048, ragged=True),
    "ragged4096_bs128": dict(S=128, H=32, D=128, B=16, ctx=4096, ragged=True),
    # long-tailed lengths, like a chat corpus
    "lognormal": dict(S=256, H=32, D=128, B=16, ctx=2048, ragged="lognormal"),
    "lognormal_bs128": dict(S=128, H=32, D=128, B=16, ctx=2048, ragged="lognormal"),
    "lognormal_bs64": dict(S=64, H=32, D=128, B=16, ctx=2048, ragged="lognormal"),
    "short256": dict(S=256, H=32, D=128, B=16, ctx=256, ragged=False),
    "bs2": dict(S=2, H=32, D=128, B=16, ctx=1024, ragged=False),
    "bs3": dict(S=3, H=32, D=128, B=16, ctx=1024, ragged=False),
    "bs4": dict(S=4, H=32, D=128, B=16, ctx=1024, ragged=False),
    "bs5": dict(S=5, H=32, D=128, B=16, ctx=1024, ragged=False),
    "bs6": dict(S=6, H=32, D=128, B=16, ctx=1024, ragged=False),
    "bs7": dict(S=7, H=32, D=128, B=16, ctx=1024, ragged=False),
    "bs4_512": dict(S=4, H=32, D=128, B=16, ctx=512, ragged=False),
    "bs4_768": dict(S=4, H=32, D=128, B=16, ctx=768, ragged=False),
    "bs6_512": dict(S=6, H=32, D=128, B=16, ctx=512, ragged=False),
    "bs6_768": dict(S=6, H=32, D=128, B=16, ctx=768, ragged=False),
    "bs4_ragged": dict(S=4, H=32, D=128, B=16, ctx=1024, ragged=True),
    "bs6_ragged": dict(S=6, H=32, D=128, B=16, ctx=1024, ragged=True),
    "c2_bs8": dict(S=8, H=12, D=64, B=16, ctx=1024, ragged=False),
    "c2_bs20": dict(S=20, H=12, D=64, B=16, ctx=1024, ragged=False),
    "short128": dict(S=256, H=32, D=128, B=16, ctx=128, ragged=False),
    "ctx512": dict(S=256, H=32, D=128, B=16, ctx=512, ragged=False),
    "lognormal_1k": dict(S=256, H=32, D=128, B=16, ctx=1024, ragged="lognormal"),
    "ragged512": dict(S=256, H=32, D=128, B=16, ctx=512, ragged=True),
    "bs64": dict(S=64, H=32, D=128, B=16, ctx=1024, ragged=False),
    "bs128": dict(S=128, H=32, D=128, B=16, ctx=1024, ragged=False),
    # batches between one and three residencies of rows (768 workgroups fit the chip)
    "bs12": dict(S=12, H=32, D=128, B=16, ctx=1024, ragged=False),
    "bs24": dict(S=24, H=32, D=128, B=16, ctx=1024, ragged=False),
    "bs32": dict(S=32, H=32, D=128, B=16, ctx=1024, ragged=False),
    "bs48": dict(S=48, H=32, D=128, B=16, ctx=1024, ragged=False),
    "bs16_ctx4096": dict(S=16, H=32, D=128, B=16, ctx=4096, ragged=False),
    "bs32_ctx2048": dict(S=32, H=32, D=128, B=16, ctx=2048, ragged=False),
    "bs40_ctx4096": dict(S=40, H=32, D=128, B=16, ctx=4096, ragged=False),
    "bs24_ctx4096": dict(S=24, H=32, D=128, B=16, ctx=4096, ragged=False),
    "bs32_ctx4096": dict(S=32, H=32, D=128, B=16, ctx=4096, ragged=False),
    "bs64_ctx4096": dict(S=64, H=32, D=128, B=16, ctx=4096, ragged=False),
    "bs48_ctx2048": dict(S=48, H=32, D=128, B=16, ctx=2048, ragged=False),
    "bs64_ctx2048": dict(S=64, H=32, D=128, B=16, ctx=2048, ragged=False),
    "bs128_ctx2048": dict(S=128, H=32, D=128, B=16, ctx=2048, ragged=False),
    "bs32_ctx8192": dict(S=32, H=32, D=128, B=16, ctx=8192, ragged=False),
    "bs40_ctx4096_ragged": dict(S=40, H=32, D=128, B=16, ctx=4096, ragged=True),
}


def load(name):
    path = os.path.join(PKG, "libpaged_attn_mi355x.so") if name == "default" else os.path.join(PKG, "tune", f"libpa_{name}.so")
    lib = ctypes.CDLL(path)
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    lib.pa_single_query_cached_kv_attention.argtypes = [vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, vp]
    lib.pa_single_query_cached_kv_attention.restype = i32
    lib.pa_last_error.restype = ctypes.c_char_p
    if hasattr(lib, "pa_single_query_cached_kv_attention_planned"):
        lib.pa_single_query_cached_kv_attention_planned.argtypes = [
            vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, vp, ctypes.c_size_t, i32, i64, vp]
        lib.pa_single_query_cached_kv_attention_planned.restype = i32
        lib.pa_decode_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
        lib.pa_decode_workspace_bytes.restype = ctypes.c_size_t
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="default")
    ap.add_argument("--shapes", default="north_star,c2,ragged")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--per-round", type=int, default=5)
    ap.add_argument("--sequential", action="store_true")
    ap.add_argument("--planned", action="store_true",
                    help="the _planned entry point with its workspace; the plan is made once, outside the timing")
    args = ap.parse_args()
    names = args.libs.split(",")
    libs = {n: load(n) for n in names}
    dev = torch.device("cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for shape in args.shapes.split(","):
        c = SHAPES[shape]
        S, H, D, B, ctx = c["S"], c["H"], c["D"], c["B"], c["ctx"]
        g = torch.Generator(device="cpu")
        g.manual_seed(0)
        ctxs = torch.randint(1, ctx + 1, (S,), generator=g).tolist() if c["ragged"] else [ctx] * S
        if c["ragged"] == "lognormal":
            import math
            import numpy as np
            ctxs = [int(min(max(v, 1), ctx)) for v in np.random.default_rng(0).lognormal(math.log(300.0), 1.0, S)]
        bps = (ctx + B - 1) // B
        share = c.get("share", 1)
        groups = (S + share - 1) // share
        half = bps // 2 if share > 1 else 0
        NB = groups * half + S * (bps - half)
        ws = 2 * NB * H * D * B * 2
        copies = max(1, min(8, int((1.5 * 2**30) // ws) + 1)) if ws < 2**30 else 1
        kcs = [torch.empty(NB, H, D // 8, B, 8, dtype=torch.float16, device=dev).normal_(0, 0.5) for _ in range(copies)]
        vcs = [torch.empty(NB, H, D, B, dtype=torch.float16, device=dev).normal_(0, 0.5) for _ in range(copies)]
        perm = torch.arange(NB, dtype=torch.int32) if args.sequential else torch.randperm(NB, generator=g).to(torch.int32)
        bt = torch.zeros(S, bps, dtype=torch.int32)
        own = perm[groups * half:].reshape(S, bps - half)
        for s_ in range(S):
            if half:
                bt[s_, :half] = perm[:groups * half].reshape(groups, half)[s_ // share]
            bt[s_, half:] = own[s_]
        bt = bt.to(dev)
        cl = torch.tensor(ctxs, dtype=torch.int32, device=dev)
        qkv = torch.empty(S, 3, H, D, dtype=torch.float16, device=dev).normal_(0, 0.5)
        q = qkv[:, 0]
        out = torch.empty(S, H, D, dtype=torch.float16, device=dev)
        nbytes = algorithmic_bytes(ctxs, H, D, B, 2)

        wss = {}
        if args.planned:
            for n in names:
                nb = libs[n].pa_decode_workspace_bytes(S, H, D, B, ctx)
                wss[n] = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
                rc = libs[n].pa_single_query_cached_kv_attention_planned(
                    out.data_ptr(), q.data_ptr(), kcs[0].data_ptr(), vcs[0].data_ptr(), 1.0 / D ** 0.5, bt.data_ptr(),
                    cl.data_ptr(), S, H, D, B, bps, ctx, q.stride(0), 1, wss[n].data_ptr(), nb, 0, sum(ctxs), stream)
                assert rc == 0, libs[n].pa_last_error()

        def call(lib, i):
            if args.planned:
                n = [k_ for k_, v_ in libs.items() if v_ is lib][0]
                rc = lib.pa_single_query_cached_kv_attention_planned(
                    out.data_ptr(), q.data_ptr(), kcs[i % copies].data_ptr(), vcs[i % copies].data_ptr(),
                    1.0 / D ** 0.5, bt.data_ptr(), cl.data_ptr(), S, H, D, B, bps, ctx, q.stride(0), 1,
                    wss[n].data_ptr(), wss[n].numel(), 1, sum(ctxs), stream)
                assert rc == 0, lib.pa_last_error()
                return
            rc = lib.pa_single_query_cached_kv_attention(
                out.data_ptr(), q.data_ptr(), kcs[i % copies].data_ptr(), vcs[i % copies].data_ptr(),
                1.0 / D ** 0.5, bt.data_ptr(), cl.data_ptr(), S, H, D, B, bps, ctx, q.stride(0), 1, stream)
            assert rc == 0, lib.pa_last_error()

        times = {n: [] for n in names}
        for n in names:
            call(libs[n], 0)
        torch.cuda.synchronize()
        k = 0
        for _ in range(args.rounds):
            for n in names:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _i in range(args.per_round):
                    call(libs[n], k)
                    k += 1
                b.record()
                torch.cuda.synchronize()
                times[n].append(a.elapsed_time(b) / args.per_round * 1e3)
        for n in names:
            t = sorted(times[n])
            med = t[len(t) // 2]
            print(f"{shape:11s} {n:12s} median {med:8.1f} us  min {t[0]:8.1f} us  "
                  f"{nbytes / med / 1e3:7.0f} GB/s ({100 * nbytes / med / 1e3 / 8000:5.1f}% of 8 TB/s)", flush=True)
        del kcs, vcs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
