#!/usr/bin/env bash
# Host-side UndefinedBehaviorSanitizer pass over the C-ABI library (no GPU needed): builds the library with the
# host half instrumented (the device half is not: the sanitizer does not exist for amdgcn), then drives the host
# logic that runs before any launch — workspace sizing / partition planning over a grid of shapes up to absurd
# sizes, and the argument checks of the entry points.  Any report aborts the run.
set -euo pipefail
cd "$(dirname "$0")/../vllm-v0.1.0_amd/csrc"
OUT=${1:-/tmp/pa_ubsan}
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
RT=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
SAN="-Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined"
COMMON="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fno-gpu-rdc -I../../include -I."
for f in pa_decode pa_prefill pa_comm; do $HIPCC $COMMON -ffast-math -fno-finite-math-only $SAN -c $f.hip -o "$OUT/$f.o" 2>/dev/null; done
for f in pa_elementwise pa_cache; do $HIPCC $COMMON -ffp-contract=off $SAN -c $f.hip -o "$OUT/$f.o" 2>/dev/null; done
$HIPCC $COMMON $SAN -x hip -c pa_error.cpp -o "$OUT/pa_error.o" 2>/dev/null
$HIPCC --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc "$OUT"/*.o "$RT/libclang_rt.ubsan_standalone-x86_64.a" \
  "$RT/libclang_rt.ubsan_standalone_cxx-x86_64.a" -o "$OUT/libpaged_attn_mi355x.so"
python3 - "$OUT/libpaged_attn_mi355x.so" <<'PY'
import ctypes, itertools, sys
lib = ctypes.CDLL(sys.argv[1])
lib.pa_decode_workspace_bytes.restype = ctypes.c_size_t
lib.pa_decode_workspace_bytes.argtypes = [ctypes.c_int] * 5
n = 0
for S, H, D, B, ctx in itertools.product((0, 1, 3, 16, 64, 255, 256, 1024, 4096, 65536), (1, 3, 12, 32, 40, 64), (64, 80, 96, 128),
                                         (8, 16, 32), (1, 15, 16, 255, 1024, 4096, 65536, 1 << 20)):
    lib.pa_decode_workspace_bytes(S, H, D, B, ctx)
    n += 1
vp, f32, i32, i64 = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_int64
one = vp(16)
lib.pa_single_query_cached_kv_attention.argtypes = [vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, vp]
bad = 0
for S, H, D, B, mb, mc, qs, dt in ((2, 2, 100, 16, 1, 16, 128, 1), (2, 2, 64, 64, 1, 16, 128, 1), (2, 2, 64, 16, 1, 32, 128, 1),
                                  (-1, 2, 64, 16, 1, 16, 128, 1), (2, 2, 64, 16, 1, 16, 128, 7), (70000, 40000, 64, 16, 1, 16, 128, 1)):
    bad += lib.pa_single_query_cached_kv_attention(one, one, one, one, 1.0, one, one, S, H, D, B, mb, mc, qs, dt, None) == 0
lib.pa_multi_query_kv_attention.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f32, i64, i64, i64, i32, vp]
cu = (ctypes.c_int32 * 3)(0, 5, 3)
bad += lib.pa_multi_query_kv_attention(one, one, one, one, cu, 2, 2, 64, 1.0, 128, 128, 128, 1, None) == 0
comm = vp()
blob = (ctypes.c_ubyte * 128)()
lib.pa_comm_create.argtypes = [i32, i32, ctypes.c_size_t, ctypes.POINTER(vp), vp]
bad += lib.pa_comm_create(3, 2, 1024, ctypes.byref(comm), blob) == 0
assert bad == 0, "an invalid call was accepted"
print(f"host UBSan pass: {n} sizing calls and the argument checks ran without a report")
PY
