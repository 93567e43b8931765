// The three small fused operators that sit next to the attention path in a
// decoder layer, for gfx950 (MI355X): rotary_embedding_neox, rms_norm,
// silu_and_mul.  They replace csrc/pos_encoding_kernels.cu,
// csrc/layernorm_kernels.cu (+ reduction_utils.cuh) and
// csrc/activation_kernels.cu of vLLM v0.1.0.
//
// All three are bandwidth-bound element-wise passes: 16-byte vector accesses
// wherever the shapes allow, wave64 reductions, one pass over HBM.
//
// Rounding follows the reference exactly, so that 16-bit results are bit-equal
// to the reference's arithmetic: its kernels compute on c10::Half / BFloat16
// values whose every operator rounds to the storage type.  This file is
// therefore built WITHOUT fast-math and with floating-point contraction off.
#include "pa_common.h"
#include "pa_exact_math.h"

namespace pa {

// ---- rotary_embedding_neox ------------------------------------------------------
//
// In place on query and key, per token and head, for i < rot_dim / 2:
//   x' = x * cos - y * sin,   y' = y * cos + x * sin,   x = v[i], y = v[i + rot_dim/2]
// cos_sin_cache row of the token's position = [cos(rot_dim/2) | sin(rot_dim/2)]
// (csrc/pos_encoding_kernels.cu:7-47).  grid = (tokens, slices of the row).
template <typename T, bool VEC>
__global__ void __launch_bounds__(256)
pa_rotary_neox_kernel(const int64_t* __restrict__ positions, T* __restrict__ query, T* __restrict__ key,
                      const T* __restrict__ cos_sin_cache, const int rot_dim, const int64_t query_stride,
                      const int64_t key_stride, const int num_heads, const int head_size) {
  constexpr int X = 16 / (int)sizeof(T);
  const int token = blockIdx.x;
  const int64_t pos = positions[token];
  const T* cache = cos_sin_cache + pos * rot_dim;
  const int embed = rot_dim / 2;
  T* qrow = query + (int64_t)token * query_stride;
  T* krow = key + (int64_t)token * key_stride;
  if (VEC) {
    typedef Pack<T, X> P;
    const int per_head = embed / X;
    const int n = num_heads * per_head;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
      const int h = i / per_head;
      const int xo = (i - h * per_head) * X;
      const P c = *reinterpret_cast<const P*>(cache + xo);
      const P s = *reinterpret_cast<const P*>(cache + embed + xo);
      T* rows[2] = {qrow, krow};
#pragma unroll
      for (int w = 0; w < 2; w++) {
        T* base = rows[w] + (int64_t)h * head_size;
        P vx = *reinterpret_cast<const P*>(base + xo);
        P vy = *reinterpret_cast<const P*>(base + embed + xo);
        P ox, oy;
#pragma unroll
        for (int e = 0; e < X; e++) {
          ox.v[e] = sub_t(mul_t(vx.v[e], c.v[e]), mul_t(vy.v[e], s.v[e]));
          oy.v[e] = add_t(mul_t(vy.v[e], c.v[e]), mul_t(vx.v[e], s.v[e]));
        }
        *reinterpret_cast<P*>(base + xo) = ox;
        *reinterpret_cast<P*>(base + embed + xo) = oy;
      }
    }
  } else {
    const int n = num_heads * embed;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
      const int h = i / embed;
      const int xo = i - h * embed;
      const T c = cache[xo], s = cache[embed + xo];
      T* rows[2] = {qrow, krow};
#pragma unroll
      for (int w = 0; w < 2; w++) {
        T* base = rows[w] + (int64_t)h * head_size;
        const T vx = base[xo], vy = base[embed + xo];
        base[xo] = sub_t(mul_t(vx, c), mul_t(vy, s));
        base[embed + xo] = add_t(mul_t(vy, c), mul_t(vx, s));
      }
    }
  }
}

// ---- rms_norm ---------------------------------------------------------------------
//
// out = ((T)(x * rsqrt(mean(x^2) + eps))) * weight   (csrc/layernorm_kernels.cu:10-34)
// One workgroup of 256 lanes per token.  The row is read once: each lane keeps
// up to KEEP 16-byte vectors in registers between the reduction and the scaling
// pass (hidden_size <= 256 * KEEP * X, i.e. 8192 for 16-bit types); longer or
// unaligned rows are re-read (they come from L2).
__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <typename T, bool VEC>
__global__ void __launch_bounds__(256)
pa_rms_norm_kernel(T* __restrict__ out, const T* __restrict__ input, const T* __restrict__ weight,
                   const float epsilon, const int hidden_size) {
  constexpr int X = 16 / (int)sizeof(T);
  constexpr int KEEP = 4;
  __shared__ float red[4];
  const int64_t row = (int64_t)blockIdx.x * hidden_size;
  float ss = 0.f;
  if (VEC) {
    typedef Pack<T, X> P;
    const int nvec = hidden_size / X;
    const P* in = reinterpret_cast<const P*>(input + row);
    P keep[KEEP];
#pragma unroll
    for (int k = 0; k < KEEP; k++) {
      const int i = threadIdx.x + 256 * k;
      if (i < nvec) {
        keep[k] = in[i];
#pragma unroll
        for (int e = 0; e < X; e++) {
          const float x = Num<T>::to(keep[k].v[e]);
          ss += x * x;
        }
      }
    }
    for (int i = threadIdx.x + 256 * KEEP; i < nvec; i += 256) {
      const P p = in[i];
#pragma unroll
      for (int e = 0; e < X; e++) {
        const float x = Num<T>::to(p.v[e]);
        ss += x * x;
      }
    }
    ss = block_sum_256(ss, red);
    const float s = 1.0f / sqrtf(ss / (float)hidden_size + epsilon);
    const P* w = reinterpret_cast<const P*>(weight);
    P* o = reinterpret_cast<P*>(out + row);
#pragma unroll
    for (int k = 0; k < KEEP; k++) {
      const int i = threadIdx.x + 256 * k;
      if (i < nvec) {
        const P wv = w[i];
        P r;
#pragma unroll
        for (int e = 0; e < X; e++) r.v[e] = mul_t(Num<T>::from(Num<T>::to(keep[k].v[e]) * s), wv.v[e]);
        o[i] = r;
      }
    }
    for (int i = threadIdx.x + 256 * KEEP; i < nvec; i += 256) {
      const P p = in[i];
      const P wv = w[i];
      P r;
#pragma unroll
      for (int e = 0; e < X; e++) r.v[e] = mul_t(Num<T>::from(Num<T>::to(p.v[e]) * s), wv.v[e]);
      o[i] = r;
    }
  } else {
    for (int i = threadIdx.x; i < hidden_size; i += 256) {
      const float x = Num<T>::to(input[row + i]);
      ss += x * x;
    }
    ss = block_sum_256(ss, red);
    const float s = 1.0f / sqrtf(ss / (float)hidden_size + epsilon);
    for (int i = threadIdx.x; i < hidden_size; i += 256)
      out[row + i] = mul_t(Num<T>::from(Num<T>::to(input[row + i]) * s), weight[i]);
  }
}

// ---- silu_and_mul -------------------------------------------------------------------
//
// out[t, i] = silu(in[t, i]) * in[t, d + i],  silu(x) = (T)(x / (1 + exp(-x)))
// (csrc/activation_kernels.cu:6-23).  grid = (tokens, slices of the row).
template <typename T> __device__ __forceinline__ T silu_t(T x) {
  const float f = Num<T>::to(x);
  return Num<T>::from(f / (1.0f + expf(-f)));
}

template <typename T, bool VEC>
__global__ void __launch_bounds__(256)
pa_silu_and_mul_kernel(T* __restrict__ out, const T* __restrict__ input, const int d) {
  constexpr int X = 16 / (int)sizeof(T);
  const int64_t token = blockIdx.x;
  const T* x = input + token * 2 * d;
  const T* y = x + d;
  T* o = out + token * d;
  if (VEC) {
    typedef Pack<T, X> P;
    const int nvec = d / X;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < nvec; i += gridDim.y * blockDim.x) {
      const P xv = reinterpret_cast<const P*>(x)[i];
      const P yv = reinterpret_cast<const P*>(y)[i];
      P r;
#pragma unroll
      for (int e = 0; e < X; e++) r.v[e] = mul_t(silu_t(xv.v[e]), yv.v[e]);
      reinterpret_cast<P*>(o)[i] = r;
    }
  } else {
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < d; i += gridDim.y * blockDim.x)
      o[i] = mul_t(silu_t(x[i]), y[i]);
  }
}

// slices of a row per token so that small batches still fill the chip
static int row_slices(int num_tokens, int work_items, int threads) {
  int slices = (work_items + threads - 1) / threads;
  const int want = (2048 + num_tokens - 1) / num_tokens;
  if (slices > want) slices = want;
  if (slices < 1) slices = 1;
  if (slices > 65535) slices = 65535;
  return slices;
}

template <typename T>
static int rotary_launch(const int64_t* positions, void* query, void* key, const void* cache, int num_tokens,
                         int num_heads, int head_size, int rot_dim, int64_t qs, int64_t ks, hipStream_t st) {
  constexpr int X = 16 / (int)sizeof(T);
  const int embed = rot_dim / 2;
  const bool vec = embed % X == 0 && head_size % X == 0 && aligned16(query) && aligned16(key) && aligned16(cache) &&
                   (qs * sizeof(T)) % 16 == 0 && (ks * sizeof(T)) % 16 == 0 && (rot_dim * sizeof(T)) % 16 == 0;
  const int work = vec ? num_heads * (embed / X) : num_heads * embed;
  const int threads = work <= 64 ? 64 : (work <= 128 ? 128 : 256);
  const dim3 grid(num_tokens, row_slices(num_tokens, work, threads));
  if (vec) {
    hipLaunchKernelGGL((pa_rotary_neox_kernel<T, true>), grid, dim3(threads), 0, st, positions, static_cast<T*>(query),
                       static_cast<T*>(key), static_cast<const T*>(cache), rot_dim, qs, ks, num_heads, head_size);
  } else {
    hipLaunchKernelGGL((pa_rotary_neox_kernel<T, false>), grid, dim3(threads), 0, st, positions,
                       static_cast<T*>(query), static_cast<T*>(key), static_cast<const T*>(cache), rot_dim, qs, ks,
                       num_heads, head_size);
  }
  PA_CHECK_LAUNCH("pa_rotary_embedding_neox launch");
  return PA_OK;
}

template <typename T>
static int rms_launch(void* out, const void* input, const void* weight, float eps, int num_tokens, int hidden,
                      hipStream_t st) {
  constexpr int X = 16 / (int)sizeof(T);
  const bool vec = hidden % X == 0 && aligned16(out) && aligned16(input) && aligned16(weight);
  if (vec) {
    hipLaunchKernelGGL((pa_rms_norm_kernel<T, true>), dim3(num_tokens), dim3(256), 0, st, static_cast<T*>(out),
                       static_cast<const T*>(input), static_cast<const T*>(weight), eps, hidden);
  } else {
    hipLaunchKernelGGL((pa_rms_norm_kernel<T, false>), dim3(num_tokens), dim3(256), 0, st, static_cast<T*>(out),
                       static_cast<const T*>(input), static_cast<const T*>(weight), eps, hidden);
  }
  PA_CHECK_LAUNCH("pa_rms_norm launch");
  return PA_OK;
}

template <typename T>
static int silu_launch(void* out, const void* input, int num_tokens, int d, hipStream_t st) {
  constexpr int X = 16 / (int)sizeof(T);
  const bool vec = d % X == 0 && aligned16(out) && aligned16(input);
  const int work = vec ? d / X : d;
  const int threads = work <= 64 ? 64 : (work <= 128 ? 128 : 256);
  const dim3 grid(num_tokens, row_slices(num_tokens, work, threads));
  if (vec) {
    hipLaunchKernelGGL((pa_silu_and_mul_kernel<T, true>), grid, dim3(threads), 0, st, static_cast<T*>(out),
                       static_cast<const T*>(input), d);
  } else {
    hipLaunchKernelGGL((pa_silu_and_mul_kernel<T, false>), grid, dim3(threads), 0, st, static_cast<T*>(out),
                       static_cast<const T*>(input), d);
  }
  PA_CHECK_LAUNCH("pa_silu_and_mul launch");
  return PA_OK;
}

static bool known_dtype(int dtype) { return dtype == PA_DTYPE_F32 || dtype == PA_DTYPE_F16 || dtype == PA_DTYPE_BF16; }

}  // namespace pa

extern "C" int pa_rotary_embedding_neox(const int64_t* positions, void* query, void* key, const void* cos_sin_cache,
                                        int num_tokens, int num_heads, int head_size, int rot_dim,
                                        int64_t query_stride, int64_t key_stride, int dtype, void* stream) {
  using namespace pa;
  clear_error();
  if (!known_dtype(dtype)) return fail(PA_ERR_UNSUPPORTED, "pa_rotary_embedding_neox: unsupported data type code: %d", dtype);
  PA_REQUIRE(num_tokens >= 0 && num_heads >= 0 && head_size >= 0 && rot_dim >= 0,
             "pa_rotary_embedding_neox: negative size");
  PA_REQUIRE(rot_dim % 2 == 0 && rot_dim <= head_size, "pa_rotary_embedding_neox: rot_dim %d must be even and <= head_size %d",
             rot_dim, head_size);
  if (num_tokens == 0 || num_heads == 0 || rot_dim == 0) return PA_OK;
  PA_REQUIRE(positions && query && key && cos_sin_cache, "pa_rotary_embedding_neox: null pointer argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case PA_DTYPE_F32: return rotary_launch<float>(positions, query, key, cos_sin_cache, num_tokens, num_heads, head_size, rot_dim, query_stride, key_stride, st);
    case PA_DTYPE_F16: return rotary_launch<_Float16>(positions, query, key, cos_sin_cache, num_tokens, num_heads, head_size, rot_dim, query_stride, key_stride, st);
    default: return rotary_launch<__bf16>(positions, query, key, cos_sin_cache, num_tokens, num_heads, head_size, rot_dim, query_stride, key_stride, st);
  }
}

extern "C" int pa_rms_norm(void* out, const void* input, const void* weight, float epsilon, int num_tokens,
                           int hidden_size, int dtype, void* stream) {
  using namespace pa;
  clear_error();
  if (!known_dtype(dtype)) return fail(PA_ERR_UNSUPPORTED, "pa_rms_norm: unsupported data type code: %d", dtype);
  PA_REQUIRE(num_tokens >= 0 && hidden_size >= 0, "pa_rms_norm: negative size");
  if (num_tokens == 0 || hidden_size == 0) return PA_OK;
  PA_REQUIRE(out && input && weight, "pa_rms_norm: null pointer argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case PA_DTYPE_F32: return rms_launch<float>(out, input, weight, epsilon, num_tokens, hidden_size, st);
    case PA_DTYPE_F16: return rms_launch<_Float16>(out, input, weight, epsilon, num_tokens, hidden_size, st);
    default: return rms_launch<__bf16>(out, input, weight, epsilon, num_tokens, hidden_size, st);
  }
}

extern "C" int pa_silu_and_mul(void* out, const void* input, int num_tokens, int d, int dtype, void* stream) {
  using namespace pa;
  clear_error();
  if (!known_dtype(dtype)) return fail(PA_ERR_UNSUPPORTED, "pa_silu_and_mul: unsupported data type code: %d", dtype);
  PA_REQUIRE(num_tokens >= 0 && d >= 0, "pa_silu_and_mul: negative size");
  if (num_tokens == 0 || d == 0) return PA_OK;
  PA_REQUIRE(out && input, "pa_silu_and_mul: null pointer argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case PA_DTYPE_F32: return silu_launch<float>(out, input, num_tokens, d, st);
    case PA_DTYPE_F16: return silu_launch<_Float16>(out, input, num_tokens, d, st);
    default: return silu_launch<__bf16>(out, input, num_tokens, d, st);
  }
}
