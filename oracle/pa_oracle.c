/*
 * CPU oracle for the PagedAttention decode hot path of vLLM v0.1.0.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is a plain-C restatement of the
 * reference algorithm, used only as the checker in tests/, in
 * __graft_entry__.smoke() and as the timed CPU baseline of bench.py.  The
 * product path (the HIP library under vllm-v0.1.0_amd/csrc) never links,
 * imports or calls anything in here.
 *
 * Parity pin: checked against the reference's own pure-torch statement of
 * the path (tests/kernels/test_attention.py:14-66, run on CPU in this
 * container by oracle/gen_golden.py) through the committed fixtures in
 * tests/golden/.  The cache operations are pure copies; their index maths
 * is pinned by the reference's inline test loops
 * (tests/kernels/test_cache.py:51-56, 95-100, 138-143).
 *
 * Element types are passed as raw storage: dtype 0 = float32, 1 = IEEE
 * binary16 (uint16_t bits), 2 = bfloat16 (uint16_t bits).
 *
 * Layouts (reference csrc/attention/attention_kernels.cu:73-79):
 *   query        [num_seqs, num_heads, head_size], row stride q_stride
 *   key_cache    [num_blocks, num_heads, head_size/x, block_size, x]
 *   value_cache  [num_blocks, num_heads, head_size, block_size]
 *   block_tables [num_seqs, max_num_blocks_per_seq] int32
 *   context_lens [num_seqs] int32
 * with x = 16 / sizeof(element) (vllm/worker/cache_engine.py:48-56).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { PAO_F32 = 0, PAO_F16 = 1, PAO_BF16 = 2 };

static int pao_elem_size(int dtype) { return dtype == PAO_F32 ? 4 : 2; }

/* ---- scalar conversions (round-to-nearest-even) ---------------------- */

static float pao_half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do {
        man <<= 1;
        e++;
      } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static uint16_t pao_float_to_half(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t absx = x & 0x7fffffffu;
  if (absx >= 0x7f800000u) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u : 0));
  }
  if (absx >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (absx < 0x33000001u) { /* < 2^-25 (or == 2^-25 tie -> even = 0) */
    return (uint16_t)sign;
  }
  int32_t e = (int32_t)(absx >> 23) - 127;
  uint32_t man = (absx & 0x7fffffu) | 0x800000u;
  uint32_t shift, half_bits;
  if (e < -14) { /* subnormal result */
    shift = (uint32_t)(13 + (-14 - e));
    half_bits = 0;
  } else {
    shift = 13;
    half_bits = (uint32_t)(e + 15) << 10;
  }
  uint32_t q = man >> shift;
  uint32_t rem = man & ((1u << shift) - 1u);
  uint32_t halfway = 1u << (shift - 1);
  if (e >= -14) q &= 0x3ffu; /* drop the implicit bit for normals */
  uint32_t r = half_bits + q;
  if (rem > halfway || (rem == halfway && (r & 1u))) r++;
  return (uint16_t)(sign | r);
}

static float pao_bf16_to_float(uint16_t b) {
  uint32_t bits = (uint32_t)b << 16;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static uint16_t pao_float_to_bf16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40u);
  uint32_t lsb = (x >> 16) & 1u;
  x += 0x7fffu + lsb;
  return (uint16_t)(x >> 16);
}

static float pao_load(const void* base, int64_t idx, int dtype) {
  switch (dtype) {
    case PAO_F32:
      return ((const float*)base)[idx];
    case PAO_F16:
      return pao_half_to_float(((const uint16_t*)base)[idx]);
    default:
      return pao_bf16_to_float(((const uint16_t*)base)[idx]);
  }
}

static void pao_store(void* base, int64_t idx, int dtype, float v) {
  switch (dtype) {
    case PAO_F32:
      ((float*)base)[idx] = v;
      break;
    case PAO_F16:
      ((uint16_t*)base)[idx] = pao_float_to_half(v);
      break;
    default:
      ((uint16_t*)base)[idx] = pao_float_to_bf16(v);
      break;
  }
}

/* round a float to the storage type and back */
static float pao_round_to(float v, int dtype) {
  if (dtype == PAO_F16) return pao_half_to_float(pao_float_to_half(v));
  if (dtype == PAO_BF16) return pao_bf16_to_float(pao_float_to_bf16(v));
  return v;
}

/* exported for the tests that pin the conversions themselves */
float pao_test_half_to_float(uint16_t h) { return pao_half_to_float(h); }
uint16_t pao_test_float_to_half(float f) { return pao_float_to_half(f); }
float pao_test_bf16_to_float(uint16_t h) { return pao_bf16_to_float(h); }
uint16_t pao_test_float_to_bf16(float f) { return pao_float_to_bf16(f); }

/* ---- decode attention -------------------------------------------------
 *
 * Follows single_query_cached_kv_attention_kernel
 * (csrc/attention/attention_kernels.cu:67-314), one (seq, head) at a time:
 *   logits[t] = scale * <q, k_t>                       (:156-168, fp32)
 *   m = max_t logits[t] over t < context_len           (:171-201)
 *   e[t] = exp(logits[t] - m); s = sum e               (:204-210)
 *   p[t] = e[t] * 1/(s + 1e-6)                         (:213-216)
 *   out[d] = sum_t p[t] * v[t][d]                      (:236-254)
 *
 * mode 0 ("exact"): dot products and sums carried in double, so that the
 *   result is the mathematical value of the formula above for the given
 *   stored inputs; only the final store rounds to the element type.
 * mode 1 ("kernel arithmetic"): reproduces the rounding points of the CUDA
 *   kernel for the P.V product: fp32 logits/softmax, p rounded to the
 *   element type (:240-241), the V_VEC-wide partial product and its
 *   pairwise sum done in element-type arithmetic
 *   (attention_generic.cuh:37-40, dtype_float16.cuh:192-200, 383-389), fp32
 *   accumulation across vectors.  It bounds how far a correct
 *   implementation may sit from the CUDA kernel's own output.
 */
static void pao_decode_one(void* out, const void* q, const void* kc,
                           const void* vc, float scale, const int32_t* table,
                           int context_len, int num_heads, int head, int D,
                           int B, int dtype, int mode, int64_t q_off,
                           int64_t out_off, float* logits, double* probs) {
  const int x = 16 / pao_elem_size(dtype);
  const int64_t tile = (int64_t)D * B; /* elements per (block, head) */
  float qf[512];
  for (int d = 0; d < D; d++) qf[d] = pao_load(q, q_off + d, dtype);

  float m = -INFINITY;
  for (int t = 0; t < context_len; t++) {
    const int64_t blk = table[t / B];
    const int off = t % B;
    const int64_t kbase = (blk * num_heads + head) * tile;
    float s;
    if (mode == 0) {
      double acc = 0.0;
      for (int d = 0; d < D; d++) {
        const int64_t ki = kbase + (int64_t)(d / x) * B * x + (int64_t)off * x + d % x;
        acc += (double)qf[d] * (double)pao_load(kc, ki, dtype);
      }
      s = (float)((double)scale * acc);
    } else {
      float acc = 0.f;
      for (int d = 0; d < D; d++) {
        const int64_t ki = kbase + (int64_t)(d / x) * B * x + (int64_t)off * x + d % x;
        acc = fmaf(qf[d], pao_load(kc, ki, dtype), acc);
      }
      s = scale * acc;
    }
    logits[t] = s;
    if (s > m) m = s;
  }

  if (mode == 0) {
    double sum = 0.0;
    for (int t = 0; t < context_len; t++) {
      probs[t] = exp((double)logits[t] - (double)m);
      sum += probs[t];
    }
    const double inv = 1.0 / (sum + 1e-6);
    double acc[512];
    for (int d = 0; d < D; d++) acc[d] = 0.0;
    for (int t = 0; t < context_len; t++) {
      const int64_t vbase = ((int64_t)table[t / B] * num_heads + head) * tile + t % B;
      const double p = probs[t] * inv;
      for (int d = 0; d < D; d++) acc[d] += p * (double)pao_load(vc, vbase + (int64_t)d * B, dtype);
    }
    for (int d = 0; d < D; d++) pao_store(out, out_off + d, dtype, (float)acc[d]);
    return;
  }

  /* mode 1 */
  float sum = 0.f;
  for (int t = 0; t < context_len; t++) {
    logits[t] = expf(logits[t] - m);
    sum += logits[t];
  }
  const float inv = 1.f / (sum + 1e-6f);
  const int nblk = (context_len + B - 1) / B;
  /* positions past context_len inside the last block hold 0 (:174) */
  for (int t = context_len; t < nblk * B; t++) logits[t] = 0.f;
  for (int t = 0; t < context_len; t++) logits[t] *= inv;

  int vec = x < B ? x : B; /* V_VEC_SIZE (:220) */
  for (int d = 0; d < D; d++) {
    float acc = 0.f;
    for (int b = 0; b < nblk; b++) {
      const int64_t vbase = ((int64_t)table[b] * num_heads + head) * tile + (int64_t)d * B;
      for (int v0 = 0; v0 < B; v0 += vec) {
        if (dtype == PAO_F32) {
          /* dtype_float32.cuh: float products, sequential float sum */
          float part = 0.f;
          for (int e = 0; e < vec; e++) {
            part += logits[b * B + v0 + e] * pao_load(vc, vbase + v0 + e, dtype);
          }
          acc += part;
        } else if (dtype == PAO_BF16) {
          /* dtype_bfloat16.cuh:190-196, 373-386: bf16 products, then the
           * pair sums and their total in fp32 */
          float part = 0.f;
          for (int e = 0; e < vec; e += 2) {
            const float p0 = pao_round_to(logits[b * B + v0 + e], dtype);
            const float p1 = pao_round_to(logits[b * B + v0 + e + 1], dtype);
            const float a0 = pao_round_to(p0 * pao_load(vc, vbase + v0 + e, dtype), dtype);
            const float a1 = pao_round_to(p1 * pao_load(vc, vbase + v0 + e + 1, dtype), dtype);
            part = (e == 0) ? (a0 + a1) : part + (a0 + a1);
          }
          acc += part;
        } else {
          /* dtype_float16.cuh:192-200, 383-389: fp16 products, packed (x2)
           * fp16 adds, then lo + hi in fp32 */
          float lo = 0.f, hi = 0.f;
          for (int e = 0; e < vec; e += 2) {
            const float p0 = pao_round_to(logits[b * B + v0 + e], dtype);
            const float p1 = pao_round_to(logits[b * B + v0 + e + 1], dtype);
            const float a0 = pao_round_to(p0 * pao_load(vc, vbase + v0 + e, dtype), dtype);
            const float a1 = pao_round_to(p1 * pao_load(vc, vbase + v0 + e + 1, dtype), dtype);
            if (e == 0) {
              lo = a0;
              hi = a1;
            } else {
              lo = pao_round_to(lo + a0, dtype);
              hi = pao_round_to(hi + a1, dtype);
            }
          }
          acc += lo + hi;
        }
      }
    }
    pao_store(out, out_off + d, dtype, acc);
  }
}

int pao_single_query_cached_kv_attention(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    float scale, const int32_t* block_tables, const int32_t* context_lens,
    int num_seqs, int num_heads, int head_size, int block_size,
    int max_num_blocks_per_seq, int64_t q_stride, int dtype, int mode) {
  if (head_size > 512 || head_size <= 0 || block_size <= 0) return 1;
  if (dtype < 0 || dtype > 2) return 1;
  int max_ctx = 0;
  for (int s = 0; s < num_seqs; s++)
    if (context_lens[s] > max_ctx) max_ctx = context_lens[s];
  const int padded = ((max_ctx + block_size - 1) / block_size) * block_size + block_size;
  int status = 0;
#pragma omp parallel
  {
    float* logits = (float*)malloc(sizeof(float) * (size_t)(padded > 0 ? padded : 1));
    double* probs = (double*)malloc(sizeof(double) * (size_t)(padded > 0 ? padded : 1));
    if (!logits || !probs) {
      free(logits);
      free(probs);
#pragma omp atomic write
      status = 2;
    } else {
#pragma omp for collapse(2) schedule(dynamic, 1)
      for (int s = 0; s < num_seqs; s++) {
        for (int h = 0; h < num_heads; h++) {
          pao_decode_one(out, query, key_cache, value_cache, scale,
                         block_tables + (int64_t)s * max_num_blocks_per_seq,
                         context_lens[s], num_heads, h, head_size, block_size,
                         dtype, mode, (int64_t)s * q_stride + (int64_t)h * head_size,
                         ((int64_t)s * num_heads + h) * head_size, logits, probs);
        }
      }
      free(logits);
      free(probs);
    }
  }
  return status;
}

/* ---- cache operations (byte-exact copies) ------------------------------ */

/* reshape_and_cache_kernel, csrc/cache_kernels.cu:155-181 */
int pao_reshape_and_cache(const void* key, const void* value, void* key_cache,
                          void* value_cache, const int32_t* slot_mapping,
                          int num_tokens, int num_heads, int head_size,
                          int block_size, int64_t key_stride,
                          int64_t value_stride, int dtype) {
  const int es = pao_elem_size(dtype);
  const int x = 16 / es;
  if (head_size % x) return 1;
  const char* k = (const char*)key;
  const char* v = (const char*)value;
  char* kc = (char*)key_cache;
  char* vc = (char*)value_cache;
  for (int t = 0; t < num_tokens; t++) {
    const int64_t slot = slot_mapping[t];
    const int64_t blk = slot / block_size;
    const int64_t off = slot % block_size;
    for (int h = 0; h < num_heads; h++) {
      for (int d = 0; d < head_size; d++) {
        const int64_t src_k = (int64_t)t * key_stride + (int64_t)h * head_size + d;
        const int64_t src_v = (int64_t)t * value_stride + (int64_t)h * head_size + d;
        const int64_t dst_k =
            ((((blk * num_heads + h) * (head_size / x) + d / x) * block_size + off) * x) + d % x;
        const int64_t dst_v = ((blk * num_heads + h) * head_size + d) * block_size + off;
        memcpy(kc + dst_k * es, k + src_k * es, (size_t)es);
        memcpy(vc + dst_v * es, v + src_v * es, (size_t)es);
      }
    }
  }
  return 0;
}

/* gather_cached_kv_kernel, csrc/cache_kernels.cu:242-269 */
int pao_gather_cached_kv(void* key, void* value, const void* key_cache,
                         const void* value_cache, const int32_t* slot_mapping,
                         int num_tokens, int num_heads, int head_size,
                         int block_size, int64_t key_stride,
                         int64_t value_stride, int dtype) {
  const int es = pao_elem_size(dtype);
  const int x = 16 / es;
  if (head_size % x) return 1;
  char* k = (char*)key;
  char* v = (char*)value;
  const char* kc = (const char*)key_cache;
  const char* vc = (const char*)value_cache;
  for (int t = 0; t < num_tokens; t++) {
    const int64_t slot = slot_mapping[t];
    const int64_t blk = slot / block_size;
    const int64_t off = slot % block_size;
    for (int h = 0; h < num_heads; h++) {
      for (int d = 0; d < head_size; d++) {
        const int64_t dst_k = (int64_t)t * key_stride + (int64_t)h * head_size + d;
        const int64_t dst_v = (int64_t)t * value_stride + (int64_t)h * head_size + d;
        const int64_t src_k =
            ((((blk * num_heads + h) * (head_size / x) + d / x) * block_size + off) * x) + d % x;
        const int64_t src_v = ((blk * num_heads + h) * head_size + d) * block_size + off;
        memcpy(k + dst_k * es, kc + src_k * es, (size_t)es);
        memcpy(v + dst_v * es, vc + src_v * es, (size_t)es);
      }
    }
  }
  return 0;
}

/* copy_blocks_kernel, csrc/cache_kernels.cu:58-77: for every layer and
 * every (src, dst) pair copy one whole key block and one whole value block.
 * pairs holds 2*num_pairs block numbers (src, dst, src, dst, ...). */
int pao_copy_blocks(void* const* key_caches, void* const* value_caches,
                    int num_layers, const int64_t* pairs, int num_pairs,
                    int64_t block_bytes) {
  for (int l = 0; l < num_layers; l++) {
    char* kc = (char*)key_caches[l];
    char* vc = (char*)value_caches[l];
    for (int p = 0; p < num_pairs; p++) {
      const int64_t src = pairs[2 * p], dst = pairs[2 * p + 1];
      memmove(kc + dst * block_bytes, kc + src * block_bytes, (size_t)block_bytes);
      memmove(vc + dst * block_bytes, vc + src * block_bytes, (size_t)block_bytes);
    }
  }
  return 0;
}

/* swap_blocks, csrc/cache_kernels.cu:29-46: one whole-block copy per pair
 * between two different tensors. */
int pao_swap_blocks(const void* src, void* dst, const int64_t* pairs,
                    int num_pairs, int64_t block_bytes) {
  for (int p = 0; p < num_pairs; p++) {
    memcpy((char*)dst + pairs[2 * p + 1] * block_bytes,
           (const char*)src + pairs[2 * p] * block_bytes, (size_t)block_bytes);
  }
  return 0;
}

/* ---- element-wise operators next to the attention path ------------------
 *
 * Each follows its reference kernel's rounding points: arithmetic on 16-bit
 * types rounds to the storage type after every operation (the reference's
 * c10::Half / BFloat16 operators do), fp32 stays fp32.
 */

/* rotary_embedding_neox_kernel, csrc/pos_encoding_kernels.cu:7-47, in place */
int pao_rotary_embedding_neox(const int64_t* positions, void* query, void* key,
                              const void* cos_sin_cache, int num_tokens,
                              int num_heads, int head_size, int rot_dim,
                              int64_t query_stride, int64_t key_stride, int dtype) {
  if (rot_dim % 2 || rot_dim > head_size) return 1;
  const int embed = rot_dim / 2;
  for (int t = 0; t < num_tokens; t++) {
    const int64_t cbase = positions[t] * rot_dim;
    for (int w = 0; w < 2; w++) {
      void* buf = w == 0 ? query : key;
      const int64_t row = (int64_t)t * (w == 0 ? query_stride : key_stride);
      for (int h = 0; h < num_heads; h++) {
        for (int i = 0; i < embed; i++) {
          const float c = pao_load(cos_sin_cache, cbase + i, dtype);
          const float s = pao_load(cos_sin_cache, cbase + embed + i, dtype);
          const int64_t xi = row + (int64_t)h * head_size + i, yi = xi + embed;
          const float x = pao_load(buf, xi, dtype), y = pao_load(buf, yi, dtype);
          const float ox = pao_round_to(pao_round_to(x * c, dtype) - pao_round_to(y * s, dtype), dtype);
          const float oy = pao_round_to(pao_round_to(y * c, dtype) + pao_round_to(x * s, dtype), dtype);
          pao_store(buf, xi, dtype, ox);
          pao_store(buf, yi, dtype, oy);
        }
      }
    }
  }
  return 0;
}

/* rms_norm_kernel, csrc/layernorm_kernels.cu:10-34 (variance summed in double
 * here: the reference's fp32 block reduction has no fixed order) */
int pao_rms_norm(void* out, const void* input, const void* weight, float epsilon,
                 int num_tokens, int hidden_size, int dtype) {
  for (int t = 0; t < num_tokens; t++) {
    const int64_t row = (int64_t)t * hidden_size;
    double ss = 0.0;
    for (int i = 0; i < hidden_size; i++) {
      const double x = pao_load(input, row + i, dtype);
      ss += x * x;
    }
    const float s = (float)(1.0 / sqrt(ss / hidden_size + (double)epsilon));
    for (int i = 0; i < hidden_size; i++) {
      const float x = pao_load(input, row + i, dtype);
      const float n = pao_round_to(x * s, dtype);
      pao_store(out, row + i, dtype, n * pao_load(weight, i, dtype));
    }
  }
  return 0;
}

/* silu_and_mul_kernel, csrc/activation_kernels.cu:6-23 */
int pao_silu_and_mul(void* out, const void* input, int num_tokens, int d, int dtype) {
  for (int t = 0; t < num_tokens; t++) {
    for (int i = 0; i < d; i++) {
      const float x = pao_load(input, (int64_t)t * 2 * d + i, dtype);
      const float y = pao_load(input, (int64_t)t * 2 * d + d + i, dtype);
      const float silu = pao_round_to(x / (1.0f + expf(-x)), dtype);
      pao_store(out, (int64_t)t * d + i, dtype, silu * y);
    }
  }
  return 0;
}

/* ---- prompt attention ---------------------------------------------------
 *
 * Causal self-attention inside each prompt of a flat token batch, the
 * computation PagedAttention.multi_query_kv_attention hands to xformers with a
 * BlockDiagonalCausalMask (vllm/model_executor/layers/attention.py:44-70) and
 * that tests/kernels/test_attention.py:69-101 restates per sequence:
 *   out[i] = sum_{j <= i} softmax_j(scale * <q_i, k_j>) v_j    (i, j in one prompt)
 * Dot products, the softmax and the weighted sum are carried in double; only
 * the final store rounds to the element type.  query / key / value are
 * [T, H, D] with free row strides, out is dense, cu_seqlens has num_seqs + 1
 * entries.
 */
int pao_multi_query_kv_attention(void* out, const void* query, const void* key,
                                 const void* value, const int32_t* cu_seqlens,
                                 int num_seqs, int num_heads, int head_size,
                                 float scale, int64_t q_stride, int64_t k_stride,
                                 int64_t v_stride, int dtype) {
  if (head_size > 512 || head_size <= 0) return 1;
  if (dtype < 0 || dtype > 2) return 1;
  int max_len = 0;
  for (int s = 0; s < num_seqs; s++) {
    const int n = cu_seqlens[s + 1] - cu_seqlens[s];
    if (n < 0) return 1;
    if (n > max_len) max_len = n;
  }
  int status = 0;
  const int D = head_size;
#pragma omp parallel
  {
    double* w = (double*)malloc(sizeof(double) * (size_t)(max_len > 0 ? max_len : 1));
    if (!w) {
#pragma omp atomic write
      status = 2;
    } else {
#pragma omp for collapse(2) schedule(dynamic, 1)
      for (int s = 0; s < num_seqs; s++) {
        for (int h = 0; h < num_heads; h++) {
          const int64_t t0 = cu_seqlens[s];
          const int n = cu_seqlens[s + 1] - cu_seqlens[s];
          for (int i = 0; i < n; i++) {
            const int64_t qo = (t0 + i) * q_stride + (int64_t)h * D;
            double m = -INFINITY;
            for (int j = 0; j <= i; j++) {
              const int64_t ko = (t0 + j) * k_stride + (int64_t)h * D;
              double acc = 0.0;
              for (int d = 0; d < D; d++)
                acc += (double)pao_load(query, qo + d, dtype) * (double)pao_load(key, ko + d, dtype);
              w[j] = (double)scale * acc;
              if (w[j] > m) m = w[j];
            }
            double sum = 0.0;
            for (int j = 0; j <= i; j++) {
              w[j] = exp(w[j] - m);
              sum += w[j];
            }
            const int64_t oo = ((t0 + i) * num_heads + h) * (int64_t)D;
            for (int d = 0; d < D; d++) {
              double acc = 0.0;
              for (int j = 0; j <= i; j++)
                acc += w[j] * (double)pao_load(value, (t0 + j) * v_stride + (int64_t)h * D + d, dtype);
              pao_store(out, oo + d, dtype, (float)(acc / sum));
            }
          }
        }
      }
      free(w);
    }
  }
  return status;
}
