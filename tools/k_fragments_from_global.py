#!/usr/bin/env python3
"""Times tools/microbench/k_fragments_from_global.hip (build first: --build, on the build host).
Prints the time one 8-wave workgroup per CU takes per 64-key K tile when every wave fetches its own MFMA
A-operand fragments from (L2-resident) global memory, next to the 2304 cycles the tile's MFMAs take per SIMD."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "vllm-v0.1.0_amd", "tune", "libk_fragments_from_global.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(ROOT, "tools", "microbench", "k_fragments_from_global.hip"), "-o", SO])
    sys.exit(0)
import torch
lib = ctypes.CDLL(SO)
vp, i32 = ctypes.c_void_p, ctypes.c_int
lib.kfrag_launch.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
out = torch.zeros(256 * 512, dtype=torch.float32, device="cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for row_elems, what in ((3 * 4096, "rows of a fused qkv projection (24 KiB apart)"), (4096, "dense [tokens, 32 heads x 128] rows (8 KiB apart)"),
                        (4096 + 64, "rows 8 KiB + 128 B apart")):
    n_tokens, iters = 2048, 8
    k = torch.randn(n_tokens, row_elems, dtype=torch.bfloat16, device="cuda:0")
    n_tiles = n_tokens // 64 * iters
    print(f"K: {n_tokens} tokens, {what}; per workgroup {n_tiles} tiles of 64 keys x 256 B, 8 waves each fetching the whole tile")
    for layout in (0, 1):
        for mfma in (0, 1):
            def call():
                assert lib.kfrag_launch(k.data_ptr(), out.data_ptr(), n_tokens, row_elems * 2, iters, layout, mfma, st) == 0
            call(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3): call()
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) / 3 * 1e3
            per_tile_ns = us * 1e3 / n_tiles
            print(f"  layout {layout} {'loads + 36 MFMAs per wave and tile' if mfma else 'loads only':36s}: {us:9.1f} us = {per_tile_ns:7.1f} ns per tile "
                  f"(~{per_tile_ns * 2.1:6.0f} cycles at 2.1 GHz; the tile's MFMAs alone: 2304 cycles per SIMD; the shipped kernel: ~4160 per tile)", flush=True)
