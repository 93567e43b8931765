#!/usr/bin/env python3
"""Measures the streaming-read ceiling of this GPU's HBM with a plain read
kernel (tools/microbench/read_bw.hip, built into vllm-v0.1.0_amd/tune/ by
`python tools/read_bw.py --build` on the build host), to put the decode kernel's
achieved GB/s next to what the memory system delivers to ANY read stream."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "vllm-v0.1.0_amd", "tune", "libread_bw.so")
if "--build" in sys.argv:
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(ROOT, "tools", "microbench", "read_bw.hip"), "-o", SO])
    sys.exit(0)
import torch
lib = ctypes.CDLL(SO)
lib.read_bw_launch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
nbytes = 4 << 30
buf = torch.empty(nbytes // 4, dtype=torch.int32, device="cuda:0").random_()
out = torch.zeros(1 << 20, dtype=torch.int32, device="cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for nt in (0, 1):
    for unroll in (4, 8, 16):
        for grid in (2048, 8192, 32768, 131072):
            def call():
                assert lib.read_bw_launch(buf.data_ptr(), nbytes, out.data_ptr(), grid, unroll, nt, st) == 0
            for _ in range(3): call()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): call()
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 10
            print(f"nt={nt} unroll={unroll:2d} grid={grid:6d}: {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)

# the loads' cache-policy bits written out (unroll 8)
lib.read_bw_policy_launch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for policy, name in enumerate(("sc0", "sc1", "sc0 sc1", "nt sc0", "nt sc1", "nt sc0 sc1")):
    for grid in (8192, 32768):
        def call():
            assert lib.read_bw_policy_launch(buf.data_ptr(), nbytes, out.data_ptr(), grid, policy, st) == 0
        for _ in range(3): call()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): call()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(f"policy {name:10s} unroll= 8 grid={grid:6d}: {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)
