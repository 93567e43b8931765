// Can the score step of the prompt-attention kernel take its K fragments straight from global memory?
//
// In pa_prompt_kernel the 8 waves of a workgroup each need the whole 64-key K tile (16 KiB at head size
// 128) as MFMA A operands: lane (r, hh) of k-step ks holds 16 bytes of key row r.  Today the tile is staged
// once in LDS and read 8 times from there.  This microbenchmark measures the alternative: every wave loads
// its fragments with global_load_dwordx4 in A-operand order, from an L2-resident K (one workgroup of 8 waves
// per CU, rows of a fused qkv projection: stride 3 x 4096 elements), with or without the tile's 36 MFMAs.
//   layout 0: k-step ks of lane (r, hh) = bytes [32 ks + 16 hh, +16) of the row   (32 contiguous bytes per row and instruction)
//   layout 1: k-step ks of lane (r, hh) = bytes [128 hh + 16 ks, +16)             (a lane walks one 128-byte line)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int LAYOUT, bool MFMA>
__global__ __launch_bounds__(512, 2) void kfrag_kernel(const unsigned char* k, float* out, int n_tokens, int row_bytes, int iters) {
  extern __shared__ unsigned char pad[];
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const int head = blockIdx.x & 31;
  const unsigned char* base = k + head * 256;
  f32x16 s0, s1, o0;
  for (int i = 0; i < 16; i++) s0[i] = s1[i] = o0[i] = 0.f;
  u32x4 x = {0u, 0u, 0u, 0u};
  bf16x8 q;
  for (int i = 0; i < 8; i++) q[i] = (__bf16)(1.0f + i);
  const int n_tiles = n_tokens / 64;
  for (int it = 0; it < iters; ++it) {
    for (int kt = 0; kt < n_tiles; ++kt) {
      u32x4 kf[16];
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        const int kb = st >> 3, ks = st & 7;
        const int off = LAYOUT == 0 ? 32 * ks + 16 * hh : 128 * hh + 16 * ks;
        kf[st] = *(const u32x4*)(base + (size_t)(kt * 64 + 32 * kb + r) * row_bytes + off);
      }
      if (MFMA) {
#pragma unroll
        for (int st = 0; st < 16; ++st) {
          if (st < 8) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[st]), q, s0, 0, 0, 0);
          else s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[st]), q, s1, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 20; ++j) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q, q, o0, 0, 0, 0);  // the tile's V^T P^T and row-sum MFMAs
      } else {
#pragma unroll
        for (int st = 0; st < 16; ++st) {
          asm volatile("" : : "v"(kf[st]));  // the fragment has to arrive
          x ^= kf[st];
        }
      }
      asm volatile("" ::: "memory");  // the next tile's (and the next pass's) loads are not this tile's
    }
  }
  float acc = (float)(x[0] ^ x[1] ^ x[2] ^ x[3]);
  for (int i = 0; i < 16; i++) acc += s0[i] + s1[i] + o0[i];
  if (acc == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = acc + pad[threadIdx.x];
}

extern "C" int kfrag_launch(const void* k, float* out, int n_tokens, int row_bytes, int iters, int layout, int mfma, void* stream) {
  const void* fns[4] = {(const void*)kfrag_kernel<0, false>, (const void*)kfrag_kernel<0, true>, (const void*)kfrag_kernel<1, false>,
                        (const void*)kfrag_kernel<1, true>};
  const void* fn = fns[layout * 2 + mfma];
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return 1;
  void* args[] = {&k, &out, &n_tokens, &row_bytes, &iters};
  if (hipLaunchKernel(fn, dim3(256), dim3(512), args, 96 * 1024, (hipStream_t)stream) != hipSuccess) return 2;
  return 0;
}
