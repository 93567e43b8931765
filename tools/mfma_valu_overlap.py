#!/usr/bin/env python3
"""Times tools/microbench/mfma_valu_overlap.hip: per SIMD, one wave's MFMA chain and another wave's
VALU mix alone, together, and interleaved in one wave (build first: --build, on the build host)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "vllm-v0.1.0_amd", "tune", "libmfma_valu_overlap.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(ROOT, "tools", "microbench", "mfma_valu_overlap.hip"), "-o", SO])
    sys.exit(0)
import torch
lib = ctypes.CDLL(SO)
lib.overlap_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(256 * 512, dtype=torch.float32, device="cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ITERS = 20000
NAMES = {0: "idle", 1: "MFMA", 2: "VALU", 3: "MFMA+VALU in one wave"}
base = {}
for lo, hi in ((1, 0), (2, 0), (1, 1), (2, 2), (1, 2), (3, 0), (3, 3)):
    def call():
        assert lib.overlap_launch(out.data_ptr(), lo, hi, ITERS, st) == 0
    call(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): call()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 3 * 1e3
    print(f"waves 0-3: {NAMES[lo]:22s} waves 4-7: {NAMES[hi]:22s} {us:9.1f} us  = {us * 1e3 / ITERS:7.1f} ns per iteration", flush=True)
