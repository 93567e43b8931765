#!/usr/bin/env python3
"""Where a wave of the prompt-attention kernel spends its cycles: runs the instrumented build
(`make -C vllm-v0.1.0_amd/csrc variant NAME=timing EXTRA_HIPFLAGS=-DPA_PROMPT_TIMING`) on a few prompt batches and
prints the average shader cycles per wave and processed tile in each phase of the tile loop.  The time stamps
(s_memtime) wait for the wave's outstanding LDS reads, so they cost a few per cent and end every phase at a drained
LDS queue; the attribution is coarse, the ranking is what it is for."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "vllm-v0.1.0_amd", "tune", "libpa_timing.so"))
ref = ctypes.CDLL(os.path.join(ROOT, "vllm-v0.1.0_amd", "libpaged_attn_mi355x.so"))
vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
for l in (lib, ref):
    l.pa_multi_query_kv_attention.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f32, i64, i64, i64, i32, vp]
lib.pa_prompt_timing_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
H, D = 32, int(os.environ.get("HEAD_SIZE", 128))
NAMES = ["issue next tile's global loads", "scores (K fragments from LDS, MFMAs)", "mask + softmax (VALU)",
         "V^T P^T (transposed LDS reads, packing, MFMAs)", "commit next tile to LDS (waits for the global loads)", "barrier"]
for shape in sys.argv[1:] or ["1x16384", "8x2048", "1x2560", "5x512"]:
    ns, n = (int(x) for x in shape.split("x"))
    T = ns * n
    qkv = torch.randn(T, 3, H, D, dtype=torch.bfloat16, device="cuda:0")
    out = torch.empty(T, H, D, dtype=torch.bfloat16, device="cuda:0")
    cu = (ctypes.c_int32 * (ns + 1))(*[i * n for i in range(ns + 1)])
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call(l):
        assert l.pa_multi_query_kv_attention(out.data_ptr(), qkv[:, 0].data_ptr(), qkv[:, 1].data_ptr(), qkv[:, 2].data_ptr(),
                                             cu, ns, H, D, D ** -0.5, 3 * H * D, 3 * H * D, 3 * H * D, 2, st) == 0

    def time_us(l):
        call(l); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            call(l)
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / 3 * 1e3

    us_ref, us_inst = time_us(ref), time_us(lib)
    buf = (ctypes.c_ulonglong * 16)()
    assert lib.pa_prompt_timing_read(buf) == 0   # clear
    call(lib); torch.cuda.synchronize()
    assert lib.pa_prompt_timing_read(buf) == 0
    v = list(buf)
    tiles, waves = max(v[6], 1), max(v[9], 1)
    loop_iters_cycles = sum(v[0:6])
    print(f"{shape}: {us_ref:.1f} us shipped build, {us_inst:.1f} us instrumented; {waves} waves, {tiles} processed wave-tiles "
          f"({tiles / waves:.1f} per wave), {loop_iters_cycles / tiles:.0f} cycles of the loop per processed wave-tile")
    for i, name in enumerate(NAMES):
        print(f"    {name:58s} {v[i] / tiles:8.0f} cycles per processed wave-tile  {100 * v[i] / loop_iters_cycles:5.1f} %")
    print(f"    prologue (query fragments, first tile staged) {v[7] / waves:8.0f} cycles per wave; epilogue {v[8] / waves:8.0f}", flush=True)
