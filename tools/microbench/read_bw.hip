// Streaming-read ceiling of the HBM: every lane issues UNROLL independent
// 16-byte loads per iteration over a contiguous span (1 KiB per wave
// instruction), xors them and writes one word per workgroup.  Built only by
// tools/read_bw.py; not part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ p, size_t n_vec, uint32_t* __restrict__ out) {
  const size_t per_wg = (n_vec + gridDim.x - 1) / gridDim.x;
  const size_t lo = per_wg * blockIdx.x;
  const size_t hi = lo + per_wg < n_vec ? lo + per_wg : n_vec;
  u32x4 acc = {0, 0, 0, 0};
  size_t i = lo + threadIdx.x;
  for (; i + (UNROLL - 1) * 256 < hi; i += UNROLL * 256) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  for (; i < hi; i += 256) acc ^= p[i];
  uint32_t x = acc.x ^ acc.y ^ acc.z ^ acc.w;
  for (int o = 32; o; o >>= 1) x ^= __shfl_xor(x, o);
  if ((threadIdx.x & 63) == 0) atomicXor(out + blockIdx.x, x);
}

// The same loop with the loads' cache-policy bits written out (inline assembly; the compiler does not
// track these loads, so the batch is followed by an explicit wait).  POLICY: 0 "sc0", 1 "sc1", 2 "sc0 sc1",
// 3 "nt sc0", 4 "nt sc1", 5 "nt sc0 sc1".
template <int POLICY>
__device__ __forceinline__ u32x4 load_policy(const u32x4* p) {
  u32x4 v;
  if (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
  if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
  if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
  if (POLICY == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int POLICY>
__global__ __launch_bounds__(256) void read_policy_kernel(const u32x4* __restrict__ p, size_t n_vec, uint32_t* __restrict__ out) {
  constexpr int UNROLL = 8;
  const size_t per_wg = (n_vec + gridDim.x - 1) / gridDim.x;
  const size_t lo = per_wg * blockIdx.x;
  const size_t hi = lo + per_wg < n_vec ? lo + per_wg : n_vec;
  u32x4 acc = {0, 0, 0, 0};
  size_t i = lo + threadIdx.x;
  for (; i + (UNROLL - 1) * 256 < hi; i += UNROLL * 256) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = load_policy<POLICY>(p + i + u * 256);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  for (; i < hi; i += 256) acc ^= p[i];
  uint32_t x = acc.x ^ acc.y ^ acc.z ^ acc.w;
  for (int o = 32; o; o >>= 1) x ^= __shfl_xor(x, o);
  if ((threadIdx.x & 63) == 0) atomicXor(out + blockIdx.x, x);
}

extern "C" int read_bw_policy_launch(const void* p, size_t bytes, void* out, int grid, int policy, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t n = bytes / 16;
#define LP(P) read_policy_kernel<P><<<grid, 256, 0, st>>>((const u32x4*)p, n, (uint32_t*)out)
  switch (policy) {
    case 0: LP(0); break;
    case 1: LP(1); break;
    case 2: LP(2); break;
    case 3: LP(3); break;
    case 4: LP(4); break;
    default: LP(5); break;
  }
#undef LP
  return (int)hipGetLastError();
}

extern "C" int read_bw_launch(const void* p, size_t bytes, void* out, int grid, int unroll, int nt, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t n = bytes / 16;
#define L(U, N) read_kernel<U, N><<<grid, 256, 0, st>>>((const u32x4*)p, n, (uint32_t*)out)
  if (unroll == 4) { if (nt) L(4, true); else L(4, false); }
  else if (unroll == 8) { if (nt) L(8, true); else L(8, false); }
  else { if (nt) L(16, true); else L(16, false); }
#undef L
  return (int)hipGetLastError();
}
