// Decode attention over the paged KV cache for gfx950 (MI355X).
//
// Replaces single_query_cached_kv_attention of vLLM v0.1.0
// (csrc/attention/attention_kernels.cu:67-314, launcher :332-401, dispatch
// :450-469) with a kernel designed around 64-wide waves and the two facts
// that decide its speed on this chip:
//
//  * it is a pure HBM stream (1 flop per byte): every (sequence, head) reads
//    context_len * head_size elements of K and of V exactly once;
//  * in both cache layouts one (block, head) tile is CONTIGUOUS memory
//    (head_size * block_size elements), so a wave can fetch it with full
//    16-byte-per-lane loads, 1 KiB per load instruction.
//
// One workgroup handles one (sequence, head); its waves take the sequence's
// KV blocks round-robin.  Per block a wave loads the K tile and the V tile
// straight into registers (no LDS staging: nothing is shared between waves),
// computes 16/8/32 logits, and folds them into a running (max, sum,
// accumulator) triple — an online softmax — so K and V are streamed in ONE
// pass with both tiles of the next block already in flight while the current
// one is being used.  There is no per-context logits buffer, hence no LDS
// limit on the context length (the reference keeps max_context_len floats in
// shared memory, :361-365).  The waves' partial triples are merged through
// LDS at the end.
//
// Arithmetic: q.k products and all softmax / P.V work are fp32 (the reference
// rounds P to the storage type and does 8-wide partial products in it,
// :240-251); the softmax denominator is (sum + 1e-6) as in the reference
// (:213).  Offsets are 64-bit.
#include <float.h>

#include "pa_common.h"

namespace pa {

// ---- element traits ---------------------------------------------------------

// Bit casts go through by-value helpers: __builtin_bit_cast applied directly
// to an element of an ext_vector (v[i]) reads element 0 for every i.
__device__ __forceinline__ f16x2 as_f16x2(uint32_t w) { return __builtin_bit_cast(f16x2, w); }
__device__ __forceinline__ bf16x2 as_bf16x2(uint32_t w) { return __builtin_bit_cast(bf16x2, w); }

struct ElemF32 {
  typedef float storage;
  static constexpr int kX = 4;  // elements per 16-byte vector
  static __device__ __forceinline__ float dot(u32x4 a, u32x4 b, float acc) {
#pragma unroll
    for (int i = 0; i < 4; i++) acc = __builtin_fmaf(__uint_as_float(a[i]), __uint_as_float(b[i]), acc);
    return acc;
  }
  // acc + sum_e p[e] * v[e]
  static __device__ __forceinline__ float pv(const float* p, u32x4 v, float acc) {
#pragma unroll
    for (int i = 0; i < 4; i++) acc = __builtin_fmaf(p[i], __uint_as_float(v[i]), acc);
    return acc;
  }
  static __device__ __forceinline__ u32x4 mask(u32x4 v, int first_invalid) {
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = i < first_invalid ? v[i] : 0u;
    return v;
  }
  static __device__ __forceinline__ storage from_float(float f) { return f; }
  static __device__ __forceinline__ float to_float(storage s) { return s; }
};

struct ElemF16 {
  typedef _Float16 storage;
  static constexpr int kX = 8;
  static __device__ __forceinline__ float dot(u32x4 a, u32x4 b, float acc) {
#pragma unroll
    for (int i = 0; i < 4; i++)
      acc = __builtin_amdgcn_fdot2(as_f16x2(a[i]), as_f16x2(b[i]), acc, false);
    return acc;
  }
  static __device__ __forceinline__ float pv(const float* p, u32x4 v, float acc) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const f16x2 h = as_f16x2(v[i]);
      acc = __builtin_fmaf(p[2 * i], (float)h[0], acc);
      acc = __builtin_fmaf(p[2 * i + 1], (float)h[1], acc);
    }
    return acc;
  }
  static __device__ __forceinline__ u32x4 mask(u32x4 v, int first_invalid) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t keep = (2 * i < first_invalid ? 0x0000ffffu : 0u) | (2 * i + 1 < first_invalid ? 0xffff0000u : 0u);
      v[i] &= keep;
    }
    return v;
  }
  static __device__ __forceinline__ storage from_float(float f) { return (_Float16)f; }
  static __device__ __forceinline__ float to_float(storage s) { return (float)s; }
};

struct ElemBF16 {
  typedef __bf16 storage;
  static constexpr int kX = 8;
  static __device__ __forceinline__ float dot(u32x4 a, u32x4 b, float acc) {
#pragma unroll
    for (int i = 0; i < 4; i++)
      acc = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(a[i]), as_bf16x2(b[i]), acc, false);
    return acc;
  }
  static __device__ __forceinline__ float pv(const float* p, u32x4 v, float acc) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      acc = __builtin_fmaf(p[2 * i], __uint_as_float(v[i] << 16), acc);
      acc = __builtin_fmaf(p[2 * i + 1], __uint_as_float(v[i] & 0xffff0000u), acc);
    }
    return acc;
  }
  static __device__ __forceinline__ u32x4 mask(u32x4 v, int first_invalid) { return ElemF16::mask(v, first_invalid); }
  static __device__ __forceinline__ storage from_float(float f) { return (__bf16)f; }
  static __device__ __forceinline__ float to_float(storage s) { return (float)s; }
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---- kernel -----------------------------------------------------------------
//
// Lane roles inside a wave, for one (block, head) tile:
//
//  K tile [D/X][B][X]: vector index = dx * B + t.  Lane = j * B + t owns token
//  t and the slices dx = j, j + KJ, ... (KJ = 64 / B lanes per token), so one
//  load instruction covers KJ consecutive dx rows = 1 KiB of contiguous
//  memory.  The per-token dot product is finished with log2(KJ) xor-shuffles.
//
//  V tile [D][B]: a row d holds VT = B / X vectors.  Lane = r * VT + tv owns
//  token vector tv of rows d = r, r + RPL, ... (RPL = 64 / VT rows per load),
//  again 1 KiB contiguous per instruction.  The probabilities of the lane's X
//  tokens come from a per-wave LDS strip written by the K-side lanes.
template <typename E, int D, int B, int NW>
__global__ void __launch_bounds__(NW * 64)
pa_decode_kernel(typename E::storage* __restrict__ out, const typename E::storage* __restrict__ q,
                 const typename E::storage* __restrict__ k_cache, const typename E::storage* __restrict__ v_cache,
                 const float scale_log2e, const int* __restrict__ block_tables,
                 const int* __restrict__ context_lens, const int num_seqs, const int num_heads,
                 const int max_blocks_per_seq, const int64_t q_stride, const int q_vec_ok) {
  typedef typename E::storage T;
  constexpr int X = E::kX;
  constexpr int NDX = D / X;
  constexpr int KJ = 64 / B;
  constexpr int NK = (NDX + KJ - 1) / KJ;
  constexpr int VT = B / X;
  constexpr int RPL = 64 / VT;
  constexpr int NV = (D + RPL - 1) / RPL;
  constexpr int TILE_BYTES = D * B * (int)sizeof(T);
  static_assert(D % X == 0 && B % X == 0 && 64 % B == 0, "unsupported tile shape");

  __shared__ __attribute__((aligned(16))) float p_strip[NW][B];
  __shared__ float part_acc[NW][D];
  __shared__ float part_m[NW];
  __shared__ float part_l[NW];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // head-major work order: the workgroups of one XCD walk consecutive
  // sequences of the same head, so sequences that share physical blocks
  // (parallel sampling, beam search) meet in one L2.
  const unsigned work = xcd_contiguous(blockIdx.x, gridDim.x);
  const int head = work / (unsigned)num_seqs;
  const int seq = work - head * (unsigned)num_seqs;

  const int ctx = context_lens[seq];
  const int nblk = (ctx + B - 1) / B;
  const int* __restrict__ table = block_tables + (int64_t)seq * max_blocks_per_seq;

  const int t = lane % B, j = lane / B;
  const int tv = lane % VT, r = lane / VT;

  // q slices of this lane (kept in the storage type; products are fp32)
  u32x4 qv[NK];
  {
    const T* qp = q + (int64_t)seq * q_stride + (int64_t)head * D;
#pragma unroll
    for (int i = 0; i < NK; i++) {
      const int dx = j + KJ * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (dx < NDX) {
        if (q_vec_ok) {
          v = *reinterpret_cast<const u32x4*>(qp + dx * X);
        } else {
          T tmp[X];
#pragma unroll
          for (int e = 0; e < X; e++) tmp[e] = qp[dx * X + e];
          v = __builtin_bit_cast(u32x4, tmp);
        }
      }
      qv[i] = v;
    }
  }

  const char* kbase = reinterpret_cast<const char*>(k_cache) + (int64_t)head * TILE_BYTES;
  const char* vbase = reinterpret_cast<const char*>(v_cache) + (int64_t)head * TILE_BYTES;
  const int64_t block_stride = (int64_t)num_heads * TILE_BYTES;
  const int k_lane_off = (j * B + t) * 16;   // == lane * 16
  const int v_lane_off = (r * VT + tv) * 16; // == lane * 16

  struct Tile {
    u32x4 k[NK];
    u32x4 v[NV];
  };

  auto load_tile = [&](Tile& tl, int phys) {
    const char* kp = kbase + (int64_t)phys * block_stride + k_lane_off;
    const char* vp = vbase + (int64_t)phys * block_stride + v_lane_off;
#pragma unroll
    for (int i = 0; i < NK; i++) {
      if ((NDX % KJ == 0) || (j + KJ * i < NDX)) {
        tl.k[i] = *reinterpret_cast<const u32x4*>(kp + i * 1024);
      } else {
        tl.k[i] = u32x4{0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int i = 0; i < NV; i++) {
      if ((D % RPL == 0) || (r + RPL * i < D)) {
        tl.v[i] = *reinterpret_cast<const u32x4*>(vp + i * 1024);
      } else {
        tl.v[i] = u32x4{0u, 0u, 0u, 0u};
      }
    }
  };

  float m = -FLT_MAX;  // running maximum (log2 domain)
  float l = 0.f;       // running sum of this lane's own token column
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) acc[i] = 0.f;

  auto consume = [&](Tile& tl, int blk) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NK; i++) s = E::dot(qv[i], tl.k[i], s);
#pragma unroll
    for (int off = B; off < 64; off <<= 1) s += __shfl_xor(s, off);
    s *= scale_log2e;
    const int tok0 = blk * B;
    const bool valid = tok0 + t < ctx;
    s = valid ? s : -FLT_MAX;
    float bm = s;
#pragma unroll
    for (int off = 1; off < B; off <<= 1) bm = fmaxf(bm, __shfl_xor(bm, off));
    const float m_new = fmaxf(m, bm);
    const float alpha = fast_exp2(m - m_new);
    const float p = valid ? fast_exp2(s - m_new) : 0.f;
    l = l * alpha + p;
    m = m_new;

    // hand the block's probabilities from the K-side lanes to the V-side lanes
    __builtin_amdgcn_wave_barrier();
    if (j == 0) p_strip[wave][t] = p;
    __builtin_amdgcn_wave_barrier();
    float pvv[X];
#pragma unroll
    for (int e = 0; e < X; e += 4) {
      const f32x4 f = *reinterpret_cast<const f32x4*>(&p_strip[wave][tv * X + e]);
      pvv[e] = f[0];
      pvv[e + 1] = f[1];
      pvv[e + 2] = f[2];
      pvv[e + 3] = f[3];
    }
    __builtin_amdgcn_wave_barrier();

    // tokens past the context in the last block: their cache bytes are
    // arbitrary (possibly NaN patterns), so zero them instead of relying on
    // p == 0
    if (tok0 + B > ctx) {
      const int first_invalid = ctx - tok0 - tv * X;  // may be <= 0 or >= X
#pragma unroll
      for (int i = 0; i < NV; i++) tl.v[i] = E::mask(tl.v[i], first_invalid);
    }
#pragma unroll
    for (int i = 0; i < NV; i++) acc[i] = E::pv(pvv, tl.v[i], acc[i] * alpha);
  };

  // two register tiles, alternating: the loads of block n+1 are issued before
  // block n is consumed
  {
    Tile ta, tb;
    int blk = wave;
    if (blk < nblk) load_tile(ta, table[blk]);
    while (blk < nblk) {
      int nb = blk + NW;
      if (nb < nblk) load_tile(tb, table[nb]);
      consume(ta, blk);
      blk = nb;
      if (blk >= nblk) break;
      nb = blk + NW;
      if (nb < nblk) load_tile(ta, table[nb]);
      consume(tb, blk);
      blk = nb;
    }
  }

  // ---- merge the waves' partial results -------------------------------------
#pragma unroll
  for (int off = 1; off < B; off <<= 1) l += __shfl_xor(l, off);
#pragma unroll
  for (int i = 0; i < NV; i++) {
    float a = acc[i];
#pragma unroll
    for (int off = 1; off < VT; off <<= 1) a += __shfl_xor(a, off);
    const int d = r + RPL * i;
    if (tv == 0 && d < D) part_acc[wave][d] = a;
  }
  if (lane == 0) {
    part_m[wave] = m;
    part_l[wave] = l;
  }
  __syncthreads();

  T* op = out + ((int64_t)seq * num_heads + head) * D;
  for (int d = tid; d < D; d += NW * 64) {
    float mg = part_m[0];
#pragma unroll
    for (int w = 1; w < NW; w++) mg = fmaxf(mg, part_m[w]);
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const float f = fast_exp2(part_m[w] - mg);
      L = __builtin_fmaf(part_l[w], f, L);
      o = __builtin_fmaf(part_acc[w][d], f, o);
    }
    op[d] = E::from_float(o / (L + 1e-6f));
  }
}

// ---- host side ----------------------------------------------------------------

template <typename E, int D, int B>
static int launch(void* out, const void* q, const void* kc, const void* vc, float scale, const int32_t* bt,
                  const int32_t* cl, int num_seqs, int num_heads, int max_blocks, int64_t q_stride,
                  hipStream_t stream) {
  typedef typename E::storage T;
  constexpr int NW = 4;
  const int q_vec_ok = aligned16(q) && ((q_stride * (int64_t)sizeof(T)) % 16 == 0) && ((D * sizeof(T)) % 16 == 0);
  const float scale_log2e = scale * 1.4426950408889634f;
  const unsigned grid = (unsigned)num_seqs * (unsigned)num_heads;
  hipLaunchKernelGGL((pa_decode_kernel<E, D, B, NW>), dim3(grid), dim3(NW * 64), 0, stream,
                     static_cast<T*>(out), static_cast<const T*>(q), static_cast<const T*>(kc),
                     static_cast<const T*>(vc), scale_log2e, bt, cl, num_seqs, num_heads, max_blocks, q_stride,
                     q_vec_ok);
  PA_CHECK_LAUNCH("pa_decode_kernel launch");
  return PA_OK;
}

template <typename E, int D>
static int dispatch_block(int B, void* out, const void* q, const void* kc, const void* vc, float scale,
                          const int32_t* bt, const int32_t* cl, int S, int H, int mb, int64_t qs, hipStream_t st) {
  switch (B) {
    case 8: return launch<E, D, 8>(out, q, kc, vc, scale, bt, cl, S, H, mb, qs, st);
    case 16: return launch<E, D, 16>(out, q, kc, vc, scale, bt, cl, S, H, mb, qs, st);
    case 32: return launch<E, D, 32>(out, q, kc, vc, scale, bt, cl, S, H, mb, qs, st);
    default: return fail(PA_ERR_UNSUPPORTED, "unsupported block size: %d (supported: 8, 16, 32)", B);
  }
}

template <typename E>
static int dispatch_head(int D, int B, void* out, const void* q, const void* kc, const void* vc, float scale,
                         const int32_t* bt, const int32_t* cl, int S, int H, int mb, int64_t qs, hipStream_t st) {
  switch (D) {
    case 64: return dispatch_block<E, 64>(B, out, q, kc, vc, scale, bt, cl, S, H, mb, qs, st);
    case 80: return dispatch_block<E, 80>(B, out, q, kc, vc, scale, bt, cl, S, H, mb, qs, st);
    case 96: return dispatch_block<E, 96>(B, out, q, kc, vc, scale, bt, cl, S, H, mb, qs, st);
    case 128: return dispatch_block<E, 128>(B, out, q, kc, vc, scale, bt, cl, S, H, mb, qs, st);
    default: return fail(PA_ERR_UNSUPPORTED, "unsupported head size: %d (supported: 64, 80, 96, 128)", D);
  }
}

}  // namespace pa

extern "C" int pa_single_query_cached_kv_attention(
    void* out, const void* query, const void* key_cache, const void* value_cache, float scale,
    const int32_t* block_tables, const int32_t* context_lens, int num_seqs, int num_heads, int head_size,
    int block_size, int max_num_blocks_per_seq, int max_context_len, int64_t q_stride, int dtype, void* stream) {
  using namespace pa;
  clear_error();
  PA_REQUIRE(num_seqs >= 0 && num_heads >= 0, "negative num_seqs (%d) or num_heads (%d)", num_seqs, num_heads);
  if (dtype != PA_DTYPE_F32 && dtype != PA_DTYPE_F16 && dtype != PA_DTYPE_BF16)
    return fail(PA_ERR_UNSUPPORTED, "unsupported data type code: %d", dtype);
  if (block_size != 8 && block_size != 16 && block_size != 32)
    return fail(PA_ERR_UNSUPPORTED, "unsupported block size: %d (supported: 8, 16, 32)", block_size);
  if (head_size != 64 && head_size != 80 && head_size != 96 && head_size != 128)
    return fail(PA_ERR_UNSUPPORTED, "unsupported head size: %d (supported: 64, 80, 96, 128)", head_size);
  if (num_seqs == 0 || num_heads == 0) return PA_OK;
  PA_REQUIRE(out && query && key_cache && value_cache && block_tables && context_lens, "null pointer argument");
  PA_REQUIRE(max_num_blocks_per_seq >= 0 && max_context_len >= 0, "negative table width or max_context_len");
  PA_REQUIRE((int64_t)max_num_blocks_per_seq * block_size >= max_context_len,
             "block table width %d x block size %d cannot hold max_context_len %d", max_num_blocks_per_seq,
             block_size, max_context_len);
  PA_REQUIRE(aligned16(key_cache) && aligned16(value_cache), "key_cache / value_cache must be 16-byte aligned");
  PA_REQUIRE((uint64_t)num_seqs * (uint64_t)num_heads < (1ull << 31), "num_seqs * num_heads too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case PA_DTYPE_F32:
      return dispatch_head<ElemF32>(head_size, block_size, out, query, key_cache, value_cache, scale, block_tables,
                                    context_lens, num_seqs, num_heads, max_num_blocks_per_seq, q_stride, st);
    case PA_DTYPE_F16:
      return dispatch_head<ElemF16>(head_size, block_size, out, query, key_cache, value_cache, scale, block_tables,
                                    context_lens, num_seqs, num_heads, max_num_blocks_per_seq, q_stride, st);
    default:
      return dispatch_head<ElemBF16>(head_size, block_size, out, query, key_cache, value_cache, scale, block_tables,
                                     context_lens, num_seqs, num_heads, max_num_blocks_per_seq, q_stride, st);
  }
}
