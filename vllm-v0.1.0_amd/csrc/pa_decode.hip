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
#include <limits.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include <algorithm>

#include "pa_common.h"

namespace pa {

// ---- element traits ---------------------------------------------------------

#ifndef PA_DECODE_BF16_DOT2
#define PA_DECODE_BF16_DOT2 1
#endif

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
  // the probabilities of a lane's X tokens, in the form pv() consumes
  struct Probs {
    float p[4];
  };
  static __device__ __forceinline__ Probs probs(const float* p) { return Probs{{p[0], p[1], p[2], p[3]}}; }
  // acc + sum_e p[e] * v[e]
  static __device__ __forceinline__ float pv(const Probs& p, u32x4 v, float acc) {
#pragma unroll
    for (int i = 0; i < 4; i++) acc = __builtin_fmaf(p.p[i], __uint_as_float(v[i]), acc);
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
  struct Probs {
    float p[8];
  };
  static __device__ __forceinline__ Probs probs(const float* p) {
    return Probs{{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]}};
  }
  static __device__ __forceinline__ float pv(const Probs& p, u32x4 v, float acc) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const f16x2 h = as_f16x2(v[i]);
      acc = __builtin_fmaf(p.p[2 * i], (float)h[0], acc);  // v_fma_mix_f32: no separate conversion
      acc = __builtin_fmaf(p.p[2 * i + 1], (float)h[1], acc);
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
#if PA_DECODE_BF16_DOT2
  // bfloat16 has no mixed-precision FMA: widening each value costs an
  // instruction of its own.  Instead the probabilities are rounded to bfloat16
  // once per block (as the reference rounds them to the storage type,
  // csrc/attention/attention_kernels.cu:236-254) and each pair of tokens is one
  // v_dot2_f32_bf16 with fp32 accumulation.
  struct Probs {
    uint32_t w[4];
  };
  static __device__ __forceinline__ Probs probs(const float* p) {
    Probs r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const bf16x2 h = {(__bf16)p[2 * i], (__bf16)p[2 * i + 1]};
      r.w[i] = __builtin_bit_cast(uint32_t, h);
    }
    return r;
  }
  static __device__ __forceinline__ float pv(const Probs& p, u32x4 v, float acc) {
#pragma unroll
    for (int i = 0; i < 4; i++)
      acc = __builtin_amdgcn_fdot2_f32_bf16(as_bf16x2(p.w[i]), as_bf16x2(v[i]), acc, false);
    return acc;
  }
#else
  struct Probs {
    float p[8];
  };
  static __device__ __forceinline__ Probs probs(const float* p) {
    return Probs{{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]}};
  }
  static __device__ __forceinline__ float pv(const Probs& p, u32x4 v, float acc) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      acc = __builtin_fmaf(p.p[2 * i], __uint_as_float(v[i] << 16), acc);
      acc = __builtin_fmaf(p.p[2 * i + 1], __uint_as_float(v[i] & 0xffff0000u), acc);
    }
    return acc;
  }
#endif
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
//
// kModeRows : one workgroup per (sequence, head) walks the whole context and
//   writes the normalised output row.
// kModeParts: one workgroup per (sequence, head, partition) walks
//   blocks_per_part KV blocks and writes its un-normalised (max, sum,
//   accumulator) triple to a workspace; pa_decode_merge_kernel combines the
//   partitions.  Every sequence gets the same number of partitions (the empty
//   ones leave the neutral element and exit): for few sequences, where the chip
//   is otherwise idle.
// kModeList : for big batches with long contexts.  Partitions of a fixed
//   length, as many as each sequence needs (at most four).  Every row's first
//   partition has a workgroup of its own, as in kModeRows;
//   pa_decode_plan_kernel lists the (sequence, partition >= 1) pairs that
//   exist from context_lens, and the grid carries one slot per pair that COULD
//   exist, ahead of the rows: a slot past the end of the list costs one scalar
//   load.  A sequence of one partition gets its output row written directly,
//   longer ones go through the workspace and the merge kernel.  One long
//   sequence among short ones is thereby spread over several workgroups that
//   start first, instead of setting the launch time alone.  The plan kernel
//   may also find that no sequence needs a second partition (see there).
// In kModeRows and kModeList the rows are walked by batch position or, given a
// workspace, in the plan kernel's balanced order (row_table).
constexpr int kModeRows = 0, kModeParts = 1, kModeList = 2;
constexpr int kItemPartBits = 10;  // list items: (sequence << 10) | partition

struct DecodeArgs {
  void* out;
  const void* q;
  const void* k_cache;
  const void* v_cache;
  const int* block_tables;
  const int* context_lens;
  float* ws_acc;  // [num_seqs * num_heads * num_parts][D]   (PART only)
  float* ws_ml;   // [num_seqs * num_heads * num_parts][2]   (PART only)
  int64_t q_stride;
  float scale_log2e;
  int num_seqs;
  int num_heads;
  int max_blocks_per_seq;
  int q_vec_ok;
  int num_parts;
  int blocks_per_part;
  int group;  // sequences per work-order group (see the kernel)
  const int2* row_table;  // balanced order: [8][slots_per_xcd] (sequence, context length), then the stamp; null = positional order
  int slots_per_xcd;      // balanced order: a multiple of group
  int unit;               // balanced order: consecutive sequences that are kept on one XCD (divides group)
  int plain_order;        // positional order with heads fastest (few rows)
  unsigned plan_tag;      // what the plan kernel stamps on a plan made for these sizes (never 0)
  const uint32_t* items;  // kModeList: the (sequence, partition >= 1) pairs that exist, in sequence order
  unsigned* counters;     // kModeList: [0] the plan's stamp, [1] number of pairs, [2] blocks per partition in force
  unsigned list_slots;    // kModeList: workgroup ids reserved for pairs = num_seqs * (num_parts - 1) * num_heads
  const int* planned_lens;  // kModeList: the context lengths the plan was made from, [num_seqs]
  // Append form (null = off): the step's new token of every sequence, [num_seqs][num_heads][D]
  // with these row strides in elements.  Its place is the last position of the context:
  // block_tables[seq][(ctx - 1) / B], offset (ctx - 1) % B.
  const void* new_k;
  const void* new_v;
  int64_t new_k_stride;
  int64_t new_v_stride;
  int new_vec_ok;         // 16-byte loads of the new key rows are possible
};

// A launch that finds no plan for its sizes in the workspace (never made, or made for another
// batch) must not pass for a result: every output row is filled with NaN instead.  `row` runs
// over a range that covers [0, rows) in every launch form: the whole grid where the item list's
// stamp is tested, the row pass (ids behind the list's slots) where the row table's is.
template <typename E, int D, typename Args>
__device__ __forceinline__ void poison_output_row(const Args& a, unsigned row) {
  typedef typename E::storage T;
  const unsigned rows = (unsigned)a.num_seqs * (unsigned)a.num_heads;
  if (row >= rows) return;
  T* op = static_cast<T*>(a.out) + (int64_t)row * D;
  for (int d = threadIdx.x; d < D; d += blockDim.x) op[d] = E::from_float(__builtin_nanf(""));
}

// Build-time knob for A/B runs: 1 = stream every K/V tile with the
// non-temporal hint, 0 = never, 2 (default) = per block, see load_tile.
#ifndef PA_DECODE_NT
#define PA_DECODE_NT 2
#endif

template <bool STREAM>
__device__ __forceinline__ u32x4 load16(const char* p) {
  if (STREAM) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return *reinterpret_cast<const u32x4*>(p);
}

// Build-time knob for A/B runs: minimum waves per SIMD asked of the register
// allocator for the shapes whose two in-flight tiles take at most 64 VGPRs.
// Default 1 = the compiler's own choice (138 VGPRs, 3 waves per SIMD at head
// 128 / block 16); asking for 4 spills and measures 3-20 % slower.
#ifndef PA_DECODE_MIN_WAVES
#define PA_DECODE_MIN_WAVES 1
#endif
template <typename E, int D, int B>
constexpr int decode_min_waves() {
  constexpr int X = E::kX;
  constexpr int NK = (D / X + 64 / B - 1) / (64 / B);
  constexpr int NV = (D + 64 / (B / X) - 1) / (64 / (B / X));
  return 2 * (NK + NV) * 4 <= 64 ? PA_DECODE_MIN_WAVES : 1;
}

template <typename E, int D, int B, int NW, int MODE>
__global__ void __launch_bounds__(NW * 64, (decode_min_waves<E, D, B>())) pa_decode_kernel(const DecodeArgs a) {
  typedef typename E::storage T;
  constexpr bool PART = MODE != kModeRows;
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

  // Work order.  The workgroups that share an XCD (and its L2) walk a
  // contiguous range of work items ordered (group of G sequences, head,
  // sequence in group):
  //  * the G sequences of one head are neighbours in time and place, so
  //    sequences that share physical blocks (parallel sampling, beam search)
  //    read them together and meet in L2;
  //  * G * num_heads is about the number of workgroups resident on an XCD, so
  //    all heads are in flight at once.  A (block, head) tile sits at
  //    (block * num_heads + head) * TILE_BYTES: with few heads in flight and
  //    consecutive physical blocks, the loads would fall on a fraction of the
  //    HBM channels (measured: 5.1 instead of 6.7 TB/s).
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // kModeList: the first list_slots workgroup ids are slots for the listed
  // (sequence, partition >= 1) pairs, heads fastest; the rest are the rows'
  // first partitions in the usual order.  The pairs go first so that the
  // partitions of long sequences start before the bulk of the batch.
  int head_l = 0, seq_l = 0;
  bool listed = false;
  unsigned row_id = blockIdx.x, row_count = gridDim.x;
  int part = 0;
  if (MODE == kModeList) {
    // a workspace whose plan was never made, or made for other sizes
    // (both words fetched before either is looked at: one trip to memory, not two)
    const unsigned plan_stamp = a.counters[0];
    const unsigned n_items = a.counters[1] * (unsigned)a.num_heads;
    // n_items > list_slots: more pairs than the caller's total_context_len allows for — it was not the sum
    if ((plan_stamp != a.plan_tag) | (n_items > a.list_slots)) {
      poison_output_row<E, D>(a, blockIdx.x);
      return;
    }
    if (blockIdx.x < a.list_slots) {
      if (blockIdx.x >= n_items) return;  // unused slot: one scalar load, then gone
      // heads fastest: the heads of a pair spread over the XCDs, whose shares of the listed work
      // are thereby equal (an XCD walking whole pairs measured the same on random lengths)
      const unsigned u = blockIdx.x;
      const unsigned pair = a.items[u / (unsigned)a.num_heads];
      head_l = (int)(u % (unsigned)a.num_heads);
      seq_l = (int)(pair >> kItemPartBits);
      part = (int)(pair & ((1u << kItemPartBits) - 1u));
      if (seq_l >= a.num_seqs || part >= a.num_parts) return;  // never true for a list made from these context_lens
      listed = true;
    } else {
      row_id = blockIdx.x - a.list_slots;
      row_count = gridDim.x - a.list_slots;
    }
  }
  const unsigned G = a.group;
  const unsigned per_group = G * (unsigned)a.num_heads;
  // kModeList: the partition length in force is the plan kernel's decision
  const int bpp = MODE == kModeList ? (int)a.counters[2] : a.blocks_per_part;
  int head, seq, ctx;
  // Balanced order: the sequence's length as it is NOW, next to the one the plan was made from.
  // A plan reused after context_lens changed (same sizes, other contents) is the caller's error;
  // the row it would get wrong is filled with NaN instead.
  int ctx_now = 0;
  bool planned_row = false;
  if (listed) {
    head = head_l;
    seq = seq_l;
    ctx = a.context_lens[seq];
  } else if (MODE != kModeParts && a.row_table != nullptr) {
    // Balanced order.  The hardware deals workgroup ids round-robin to the 8
    // XCDs, and an XCD streams little faster when the others have run dry: the
    // launch lasts as long as the XCD with the most bytes, and as its longest
    // row if that row starts late.  pa_decode_plan_kernel therefore places the
    // sequences by length: row_table[x][i] = (sequence, its context length) of
    // the i-th sequence XCD x works on, -1 where x has fewer.  The XCD walks
    // its sequences in chunks of G, ordered (head, sequence) as below.
    const unsigned x = row_id & 7u, w = row_id >> 3;
    const unsigned c = w / per_group, in = w - c * per_group;
    const unsigned hd = in / G;
    const unsigned i = c * G + (in - hd * G);
    // The stamp and the entry are fetched together and looked at together: tested one after the
    // other (and the stamp's two words with a short-circuit), each load waited for the previous
    // one's answer — three trips to memory ahead of every row instead of one.
    const int2 stamp = a.row_table[8u * (unsigned)a.slots_per_xcd];
    int2 e = a.row_table[x * (unsigned)a.slots_per_xcd + i];
    // (the context length too: left to itself the compiler fetches that word behind the branch)
    asm volatile("" : "+v"(e.x), "+v"(e.y));
    e.x = __builtin_amdgcn_readfirstlane(e.x);
    e.y = __builtin_amdgcn_readfirstlane(e.y);
    const bool no_plan = ((unsigned)stamp.x != a.plan_tag) | ((unsigned)stamp.y != ~a.plan_tag);  // not a plan for these sizes
    const bool empty_slot = (unsigned)e.x >= (unsigned)a.num_seqs;                                  // -1
    if (no_plan | empty_slot) {
      if (no_plan) poison_output_row<E, D>(a, row_id);
      return;
    }
    head = (int)hd;
    seq = e.x;
    ctx = e.y;
    ctx_now = a.context_lens[seq];  // looked at when the row is written: off the row's chain of dependent loads
    planned_row = true;
  } else if (MODE != kModeParts && a.plain_order) {
    // Few rows (all resident at once): heads fastest, so that every XCD gets an
    // eighth of every sequence's heads whatever the lengths are.
    seq = (int)(row_id / (unsigned)a.num_heads);
    head = (int)(row_id - (unsigned)seq * (unsigned)a.num_heads);
    ctx = a.context_lens[seq];
  } else {
    unsigned work = xcd_contiguous(row_id, row_count);
    if (MODE == kModeParts) {
      part = work % (unsigned)a.num_parts;
      work /= (unsigned)a.num_parts;
    }
    const unsigned full_groups = (unsigned)a.num_seqs / G;  // a trailing partial group keeps the last place
    unsigned grp = work / per_group;
    const unsigned in_grp = work - grp * per_group;
    if (grp < full_groups) {
      const unsigned q = full_groups >> 3, rem = full_groups & 7u;  // XCD x owns q + (x < rem) groups
      const unsigned head_run = rem * (q + 1u);
      unsigned x, i;
      if (grp < head_run) {
        x = grp / (q + 1u);
        i = grp - x * (q + 1u);
      } else {
        x = rem + (grp - head_run) / q;
        i = (grp - head_run) % q;
      }
      grp = x + 8u * i;
    }
    const unsigned gsize = min(G, (unsigned)a.num_seqs - grp * G);  // the last group may be short
    head = (int)(in_grp / gsize);
    seq = (int)(grp * G + (in_grp - (in_grp / gsize) * gsize));
    ctx = a.context_lens[seq];
  }

  // never beyond the table's width (a plan made from other context_lens may say more)
  const int nblk_seq = min((ctx + B - 1) / B, a.max_blocks_per_seq);
  // kModeList: a sequence that fits one partition is finished here, without the workspace
  const bool direct = !PART || (MODE == kModeList && nblk_seq <= bpp);
  const int blk_begin = PART ? part * bpp : 0;
  // the last partition runs to the end of the context even if the context is
  // longer than the max_context_len the partition plan was made for
  const int nblk = (PART && part + 1 < a.num_parts) ? min(nblk_seq, blk_begin + bpp) : nblk_seq;
  if (MODE == kModeParts && blk_begin >= nblk) {
    // An empty partition leaves the neutral element behind (weight zero), so that the merge kernel
    // can read all the partitions of every row without fetching the context length first.
    const int64_t slot = ((int64_t)seq * a.num_heads + head) * a.num_parts + part;
    for (int d = threadIdx.x; d < D; d += NW * 64) a.ws_acc[slot * D + d] = 0.f;
    if (threadIdx.x == 0) {
      a.ws_ml[2 * slot] = -FLT_MAX;
      a.ws_ml[2 * slot + 1] = 0.f;
    }
    return;
  }
  const int* __restrict__ table = a.block_tables + (int64_t)seq * a.max_blocks_per_seq;
  // Neighbouring sequences' tables, to spot shared physical blocks (forked
  // sequences keep the parent's blocks and sit next to it in the batch).
  const int* __restrict__ table_prev = table - (seq > 0 ? a.max_blocks_per_seq : 0);
  const int* __restrict__ table_next = table + (seq + 1 < a.num_seqs ? a.max_blocks_per_seq : 0);

  const int t = lane % B, j = lane / B;
  const int tv = lane % VT, r = lane / VT;

  // q slices of this lane (kept in the storage type; products are fp32)
  u32x4 qv[NK];
  {
    const T* qp = static_cast<const T*>(a.q) + (int64_t)seq * a.q_stride + (int64_t)head * D;
#pragma unroll
    for (int i = 0; i < NK; i++) {
      const int dx = j + KJ * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (dx < NDX) {
        if (a.q_vec_ok) {
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

  const char* kbase = static_cast<const char*>(a.k_cache) + (int64_t)head * TILE_BYTES + lane * 16;
  const char* vbase = static_cast<const char*>(a.v_cache) + (int64_t)head * TILE_BYTES + lane * 16;
  const int64_t block_stride = (int64_t)a.num_heads * TILE_BYTES;

  struct Tile {
    u32x4 k[NK];
    u32x4 v[NV];
  };

  // lane l reads bytes [l*16, l*16+16) of every 1 KiB slice of the tile: for K
  // that is (dx = j + KJ*i, token t), for V (row r + RPL*i, token vector tv).
  auto load_tile_as = [&](Tile& tl, int phys, auto stream_tag) {
    constexpr bool STREAM = decltype(stream_tag)::value;
    const char* kp = kbase + (int64_t)phys * block_stride;
    const char* vp = vbase + (int64_t)phys * block_stride;
#pragma unroll
    for (int i = 0; i < NK; i++) {
      if ((NDX % KJ == 0) || (j + KJ * i < NDX)) {
        tl.k[i] = load16<STREAM>(kp + i * 1024);
      } else {
        tl.k[i] = u32x4{0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int i = 0; i < NV; i++) {
      if ((D % RPL == 0) || (r + RPL * i < D)) {
        tl.v[i] = load16<STREAM>(vp + i * 1024);
      } else {
        tl.v[i] = u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  // Load policy per block.  A block read by this sequence only is streamed
  // with the non-temporal hint: it is used once, and keeping it out of the
  // caches' retention is worth +11 % bandwidth on MI355X.  A block that a
  // neighbouring sequence also maps at the same position is loaded with the
  // default policy so that the second reader hits L2.  (Entries of a
  // neighbour's table past its own context are padding; a false match only
  // costs the hint.)
  // Block-table entries of this wave, 64 at a time, one per lane: lane i
  // holds the entry of the wave's (chunk * 64 + i)-th block, i.e. logical block
  // blk_begin + wave + NW * (chunk * 64 + i).  The loop picks its entry with
  // v_readlane, so no memory operation sits between a tile's block number and
  // its loads.  Bit 31 of an entry = "a neighbouring sequence maps the same
  // physical block at this position".
  int entries = 0;
  auto fetch_entries = [&](int it0) {
    const int b = blk_begin + wave + NW * (it0 + lane);
    int e = 0;
    if (b < nblk) {
      const int phys = table[b];
      e = phys;
#if PA_DECODE_NT == 2
      // all three entries are fetched at once (table_prev / table_next point at this sequence's
      // own table where there is no neighbour): a short-circuit here would put the second
      // neighbour's load behind the first one's arrival, a whole trip to memory in every row's chain
      const int before = table_prev[b], after = table_next[b];
      const bool shared = ((table_prev != table) & (before == phys)) | ((table_next != table) & (after == phys));
      e |= shared ? (int)0x80000000 : 0;
#endif
    }
    entries = e;
  };
  // entry of the wave's it-th block (it is wave-uniform)
  auto entry = [&](int it) -> int {
    if ((it & 63) == 0) fetch_entries(it);
    return __builtin_amdgcn_readlane(entries, it & 63);
  };
  auto load_tile = [&](Tile& tl, int e) {
    const int phys = e & 0x7fffffff;
#if PA_DECODE_NT == 1
    load_tile_as(tl, phys, std::true_type{});
#elif PA_DECODE_NT == 0
    load_tile_as(tl, phys, std::false_type{});
#else
    // The two arms differ only in the loads' cache-policy bit; the markers
    // keep the compiler from hoisting or sinking the loads into one common
    // copy without the hint.
    if (e < 0) {
      load_tile_as(tl, phys, std::false_type{});
    } else {
      asm volatile("; streamed tile" ::: "memory");
      load_tile_as(tl, phys, std::true_type{});
      asm volatile("; end of streamed tile" ::: "memory");
    }
#endif
  };

  // Append form: the wave that holds the context's last block in registers puts the new
  // token's key and value into the tile before using it, and writes them to the caches: the
  // key as the token's D / X 16-byte vectors, the value tile whole (it is in registers anyway,
  // and whole lines go straight to memory where 2-byte stores would leave partial sectors
  // behind for the next launch to trip over).  The block being appended to belongs to this
  // sequence alone (a shared last block is copied before it is written to,
  // vllm/core/block_manager.py:139-160), and within the launch only this wave reads it.
  // Measured on MI355X (32 heads x 128, 1024 tokens, per layer, against reshape_and_cache +
  // this kernel): 1 sequence 18.4 -> 11.5 us, 4: 21.3 -> 19.6, 16: 48.1 -> 45.5; but 32: 93.7 ->
  // 96.8, 128: 333 -> 399, 256: 657 -> 731: once the launch saturates HBM with reads, stores
  // trickling in all through it (a row's worth whenever a workgroup ends) cost far more than
  // the same stores in a launch of their own (35 MB: +70 us inside, 23 us apart; with the
  // stores left out the append form costs nothing: 633 vs 633 us; the separate launch run
  // concurrently on a second stream: 675 us).  The callers use this form for small launches only.
  const int last_blk = (a.new_k != nullptr && ctx > 0 && (ctx - 1) / B < nblk_seq) ? (ctx - 1) / B : -1;
  auto append_token = [&](Tile& tl) {
    const int64_t phys = table[last_blk];
    const int tok = (ctx - 1) % B;
    char* kp = const_cast<char*>(kbase) + phys * block_stride;
    char* vp = const_cast<char*>(vbase) + phys * block_stride;
    const T* nk = static_cast<const T*>(a.new_k) + (int64_t)seq * a.new_k_stride + (int64_t)head * D;
    const T* nv = static_cast<const T*>(a.new_v) + (int64_t)seq * a.new_v_stride + (int64_t)head * D;
    if (t == tok) {
#pragma unroll
      for (int i = 0; i < NK; i++) {
        const int dx = j + KJ * i;
        if (dx < NDX) {
          u32x4 v;
          if (a.new_vec_ok) {
            v = *reinterpret_cast<const u32x4*>(nk + dx * X);
          } else {
            T tmp[X];
#pragma unroll
            for (int e = 0; e < X; e++) tmp[e] = nk[dx * X + e];
            v = __builtin_bit_cast(u32x4, tmp);
          }
          tl.k[i] = v;
          __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(kp + i * 1024));
        }
      }
    }
    const int tvs = tok / X, el = tok % X;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int d = r + RPL * i;
      if (d < D) {
        u32x4 v = tl.v[i];
        if (tv == tvs) {
          const T val = nv[d];
          if constexpr (sizeof(T) == 4) {
            const uint32_t bits = __builtin_bit_cast(uint32_t, val);
#pragma unroll
            for (int w = 0; w < 4; w++) v[w] = w == el ? bits : v[w];
          } else {
            const uint32_t bits = (uint32_t)__builtin_bit_cast(uint16_t, val);
            const int sh = (el & 1) * 16;
#pragma unroll
            for (int w = 0; w < 4; w++) v[w] = w == (el >> 1) ? ((v[w] & ~(0xffffu << sh)) | (bits << sh)) : v[w];
          }
          tl.v[i] = v;
        }
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(vp + i * 1024));
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
    s *= a.scale_log2e;
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
    const typename E::Probs probs = E::probs(pvv);
#pragma unroll
    for (int i = 0; i < NV; i++) acc[i] = E::pv(probs, tl.v[i], acc[i] * alpha);
  };

  // Two register tiles, alternating: the loads of block n+1 are issued before
  // block n is consumed.
  // Append form: the loop fetches the context's last block like any other but leaves it to
  // the code behind the loop, which holds the stores (stores inside the loop would make every
  // wait for a tile wait for them as well: loads and stores share one counter).
  {
    Tile ta, tb;
    const bool last_here = last_blk >= blk_begin && last_blk < nblk;  // in this workgroup's partition
    const int nblk_loop = last_here ? last_blk : nblk;
    int blk = blk_begin + wave;
    int it = 0;  // index of blk among this wave's blocks
    if (blk < nblk) load_tile(ta, entry(0));
    while (blk < nblk_loop) {
      int nb = blk + NW;
      if (nb < nblk) load_tile(tb, entry(it + 1));
      consume(ta, blk);
      blk = nb;
      it++;
      if (blk >= nblk_loop) break;
      nb = blk + NW;
      if (nb < nblk) load_tile(ta, entry(it + 1));
      consume(tb, blk);
      blk = nb;
      it++;
    }
    if (last_here && blk == last_blk) {  // this wave's turn came to the last block: its tile is the one just fetched
      if (it & 1) {
        append_token(tb);
        consume(tb, blk);
      } else {
        append_token(ta);
        consume(ta, blk);
      }
    }
  }

  // ---- merge the waves' partial results -------------------------------------
#pragma unroll
  for (int off = 1; off < B; off <<= 1) l += __shfl_xor(l, off);
#pragma unroll
  for (int i = 0; i < NV; i++) {
    float v = acc[i];
#pragma unroll
    for (int off = 1; off < VT; off <<= 1) v += __shfl_xor(v, off);
    const int d = r + RPL * i;
    if (tv == 0 && d < D) part_acc[wave][d] = v;
  }
  if (lane == 0) {
    part_m[wave] = m;
    part_l[wave] = l;
  }
  __syncthreads();

  float mg = part_m[0];
#pragma unroll
  for (int w = 1; w < NW; w++) mg = fmaxf(mg, part_m[w]);
  float f[NW];
  float L = 0.f;
#pragma unroll
  for (int w = 0; w < NW; w++) {
    f[w] = fast_exp2(part_m[w] - mg);
    L = __builtin_fmaf(part_l[w], f[w], L);
  }
  const int64_t row = (int64_t)seq * a.num_heads + head;
  if (!direct) {
    const int64_t slot = row * a.num_parts + part;
    for (int d = tid; d < D; d += NW * 64) {
      float o = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w++) o = __builtin_fmaf(part_acc[w][d], f[w], o);
      a.ws_acc[slot * D + d] = o;
    }
    if (tid == 0) {
      a.ws_ml[2 * slot] = mg;
      a.ws_ml[2 * slot + 1] = L;
    }
  } else {
    T* op = static_cast<T*>(a.out) + row * D;
    // (kModeList: rows that go through the workspace, and these too, are checked by the merge kernel)
    const float inv = (planned_row && ctx_now != ctx) ? __builtin_nanf("") : 1.f / (L + 1e-6f);
    for (int d = tid; d < D; d += NW * 64) {
      float o = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w++) o = __builtin_fmaf(part_acc[w][d], f[w], o);
      op[d] = E::from_float(o * inv);
    }
  }
}

// Plans one launch from context_lens: one workgroup of kPlanThreads.
//
// 1. (item-list form) Whether to split at all (also not when all rows are resident at once and
//    the longest is within a quarter of the mean length: nothing to even out).  Partitions exist to keep one
//    long sequence from setting the launch time alone.  A row workgroup shares
//    the GPU with resident_slots - 1 others, so the longest row, which starts
//    first, ends in time as long as it is no longer than the average work per
//    slot, blocks * heads / resident_slots; splitting then only adds a second
//    workgroup per row, the workspace round trip and the merge (3-4 % at 256
//    sequences of up to 2048 tokens).  The rule: split when the longest
//    sequence exceeds split_pct percent of that average.  The partition length
//    in force goes to counters[2] for the decode and merge kernels: the planned
//    one, or kNoSplitBlocks.
// 2. (item-list form) The list of (sequence, partition >= 1) pairs.  A
//    sequence has ceil(blocks / blocks_per_part) partitions, at most max_parts
//    (the last partition then runs to the end of the context).
// 3. (balanced order) row_table.  Units of `unit` consecutive sequences
//    (neighbours in the batch may share physical blocks and should meet in one
//    L2) are ranked by the blocks their row workgroups will read, heaviest
//    first, and dealt to the 8 XCDs in a snake (ranks 0..7 -> XCD 0..7,
//    8..15 -> 7..0, ...), which evens out the XCDs' totals; each XCD then takes
//    its sequences longest first, so that the launch ends on its shortest
//    rows.  Ranks come from counting (keys are distinct): no sort needed for
//    at most kMaxBalancedSeqs sequences.
constexpr int kCUs = 256;  // compute units of the MI355X
constexpr int kNoSplitBlocks = 1 << 24;
constexpr int kPlanThreads = 1024;
constexpr int kMaxBalancedSeqs = 1024;  // = kPlanThreads: one sequence per thread
constexpr int kMaxTableSlots = 8 * (kMaxBalancedSeqs / 8 + 32);

struct PlanArgs {
  const int* context_lens;
  int num_seqs;
  int block_size;
  int blocks_per_part;  // planned partition length (item-list form)
  int max_parts;        // 1 = no item list
  int split_pct;
  int num_heads;
  int resident_slots;   // workgroups of the decode kernel the GPU holds at a time
  int keep_parts_min_blocks;  // a longest context of this many blocks keeps its partitions whatever the others are (few_rows_min_blocks)
  uint32_t* items;
  unsigned* counters;
  int* planned_lens;    // item-list form: a copy of context_lens, for the merge kernel to compare with
  int2* row_table;      // null = no balanced order
  int unit;
  int slots_per_xcd;
  unsigned tag;         // stamped on the item list (counters[0]) and behind the row table
};

// number of keys in [lo, lo + span) among keys[0 .. n), n a multiple of 4 (padding keys are 0 and lo > 0)
__device__ __forceinline__ unsigned count_in_range(const unsigned* keys, int n, unsigned lo, unsigned span) {
  unsigned c = 0;
#pragma unroll 4
  for (int t = 0; t < n; t += 4) {
    const u32x4 k = *reinterpret_cast<const u32x4*>(keys + t);
    c += (k.x - lo < span ? 1u : 0u) + (k.y - lo < span ? 1u : 0u) + (k.z - lo < span ? 1u : 0u) +
         (k.w - lo < span ? 1u : 0u);
  }
  return c;
}

__global__ void __launch_bounds__(kPlanThreads) pa_decode_plan_kernel(const PlanArgs p) {
  __shared__ unsigned scan[kPlanThreads];
  __shared__ unsigned long long sums[kPlanThreads];
  __shared__ unsigned base;
  __shared__ __attribute__((aligned(16))) unsigned seq_key[kMaxBalancedSeqs + 4];
  __shared__ __attribute__((aligned(16))) unsigned unit_key[kMaxBalancedSeqs + 4];
  __shared__ __attribute__((aligned(16))) unsigned tail_key[kMaxBalancedSeqs + 4];
  __shared__ int table[kMaxTableSlots];
  const int tid = threadIdx.x, nthreads = blockDim.x;  // nthreads: a power of two
  const int num_seqs = p.num_seqs, block_size = p.block_size;
  int bpp = kNoSplitBlocks;
  if (p.max_parts > 1) {
    unsigned long long total = 0;
    unsigned longest = 0;
    for (int seq = tid; seq < num_seqs; seq += nthreads) {
      const int len = p.context_lens[seq];
      p.planned_lens[seq] = len;
      const unsigned nblk = (unsigned)((max(len, 0) + block_size - 1) / block_size);
      total += nblk;
      longest = max(longest, nblk);
    }
    sums[tid] = total;
    scan[tid] = longest;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int off = nthreads / 2; off >= 1; off >>= 1) {
      if (tid < off) {
        sums[tid] += sums[tid + off];
        scan[tid] = max(scan[tid], scan[tid + off]);
      }
      __syncthreads();
    }
    bool fits = (unsigned long long)scan[0] * 100ull * (unsigned)p.resident_slots <=
                sums[0] * (unsigned long long)p.num_heads * (unsigned long long)p.split_pct;
    // every row resident at once and the longest within a quarter of the mean: the rows end together
    // anyway (unless the rows leave half the CUs idle and are long: see few_rows_min_blocks)
    fits = fits || ((unsigned long long)num_seqs * (unsigned)p.num_heads <= (unsigned)p.resident_slots &&
                    scan[0] < (unsigned)p.keep_parts_min_blocks &&
                    (unsigned long long)scan[0] * (unsigned)num_seqs * 100ull <= sums[0] * 125ull);
    bpp = fits ? kNoSplitBlocks : p.blocks_per_part;
    __syncthreads();
    for (int s0 = 0; s0 < num_seqs; s0 += nthreads) {
      const int seq = s0 + tid;
      unsigned extra = 0;
      if (seq < num_seqs) {
        const int nblk = (max(p.context_lens[seq], 0) + block_size - 1) / block_size;
        extra = (unsigned)(min(max((nblk + bpp - 1) / bpp, 1), p.max_parts) - 1);
      }
      scan[tid] = extra;
      __syncthreads();
      for (int off = 1; off < nthreads; off <<= 1) {  // inclusive scan
        const unsigned add = tid >= off ? scan[tid - off] : 0u;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
      }
      const unsigned first = base + scan[tid] - extra;
      for (unsigned q = 0; q < extra; q++) p.items[first + q] = ((uint32_t)seq << kItemPartBits) | (q + 1u);
      __syncthreads();
      if (tid == nthreads - 1) base += scan[nthreads - 1];
      __syncthreads();
    }
    if (tid == 0) {
      p.counters[0] = p.tag;
      p.counters[1] = base;
      p.counters[2] = (unsigned)bpp;
    }
  }
  if (p.row_table == nullptr) return;

  // ---- balanced order (num_seqs <= kMaxBalancedSeqs <= nthreads... one sequence per thread) ----
  // Keys: [31:29] XCD (sequence keys only, filled in below), [28:10] weight in
  // blocks, [9:0] 1023 - index.  Distinct and non-zero; ties go to batch order.
  const int U = p.unit, NX = p.slots_per_xcd;
  const int n_units = (num_seqs + U - 1) / U;
  int my_ctx = 0;
  unsigned my_weight = 0;
  if (tid < num_seqs) {
    my_ctx = p.context_lens[tid];
    my_weight = (unsigned)min((max(my_ctx, 0) + block_size - 1) / block_size, min(bpp, (1 << 17) - 1));
    scan[tid] = my_weight;
  }
  if (tid < 4) {  // padding of the last vector of four keys
    seq_key[((num_seqs + 3) & ~3) - 1 - tid] = 0u;
    tail_key[((num_seqs + 3) & ~3) - 1 - tid] = 0u;
    unit_key[((n_units + 3) & ~3) - 1 - tid] = 0u;
  }
  for (int i = tid; i < 8 * NX; i += nthreads) table[i] = -1;
  __syncthreads();
  if (tid < n_units) {
    unsigned weight = 0;
    for (int e = 0; e < U; e++)
      if (tid * U + e < num_seqs) weight += scan[tid * U + e];
    unit_key[tid] = (min(weight, (1u << 19) - 1u) << 10) | (unsigned)(kMaxBalancedSeqs - 1 - tid);
  }
  __syncthreads();
  // Whole rows of 8 ranks are dealt as units.  The units of the last, partial
  // row (the lightest ones) are taken apart and their sequences dealt one by
  // one, so that the XCDs' sequence counts differ by one at most, not by a unit.
  const unsigned full_rows = (unsigned)n_units >> 3;
  if (tid < n_units) {
    const unsigned key = unit_key[tid];
    const unsigned rank = count_in_range(unit_key, (n_units + 3) & ~3, key + 1u, ~key);  // keys above this one
    const unsigned row = rank >> 3, col = rank & 7u;
    const unsigned x = (row & 1u) ? 7u - col : col;
    for (int e = 0; e < U; e++) {
      const int sq = tid * U + e;
      if (sq >= num_seqs) break;
      const unsigned own = (scan[sq] << 10) | (unsigned)(kMaxBalancedSeqs - 1 - sq);
      seq_key[sq] = row < full_rows ? (x << 29) | own : 0u;
      tail_key[sq] = row < full_rows ? 0u : own;
    }
  }
  __syncthreads();
  if (tid < num_seqs && tail_key[tid] != 0u) {
    const unsigned key = tail_key[tid];
    const unsigned rank = count_in_range(tail_key, (num_seqs + 3) & ~3, key + 1u, ~key);
    const unsigned row = full_rows + (rank >> 3), col = rank & 7u;
    seq_key[tid] = (((row & 1u) ? 7u - col : col) << 29) | key;
  }
  __syncthreads();
  if (tid < num_seqs) {
    const unsigned key = seq_key[tid];
    const unsigned top = (key | 0x1fffffffu);  // the largest key of the same XCD
    const unsigned place = count_in_range(seq_key, (num_seqs + 3) & ~3, key + 1u, top - key);  // same XCD, above
    table[(key >> 29) * NX + place] = tid;  // place < ceil(units / 8) * U <= NX
  }
  __syncthreads();
  if (tid < num_seqs) scan[tid] = (unsigned)my_ctx;
  __syncthreads();
  for (int i = tid; i < 8 * NX; i += nthreads) {
    const int sq = table[i];
    p.row_table[i] = make_int2(sq, sq >= 0 ? (int)scan[sq] : 0);
  }
  if (tid == 0) p.row_table[8 * NX] = make_int2((int)p.tag, (int)~p.tag);
}

// Combines the partitions of one (sequence, head): one wave per output row, lanes along d.
constexpr int kMergeGroup = 16;  // accumulator rows in flight per lane
template <typename E, int D>
__global__ void __launch_bounds__(64)
pa_decode_merge_kernel(const DecodeArgs a, const int block_size, const int skip_single) {
  typedef typename E::storage T;
  constexpr int PER = (D + 63) / 64;
  const int lane = threadIdx.x;
  const int64_t row = blockIdx.x;  // seq * num_heads + head
  // Uniform partitions: every partition of every row was written (the empty ones hold the neutral
  // element), so nothing has to be known about the row before its partitions are fetched.
  int nparts = a.num_parts;
  if (skip_single) {
    // kModeList: the partition length is the plan kernel's, and only the partitions that exist were written.
    // No plan for these sizes in the workspace, or more listed pairs than slots: the decode kernel poisoned the output
    if (a.counters[0] != a.plan_tag || a.counters[1] * (unsigned)a.num_heads > a.list_slots) return;
    if (a.row_table != nullptr) {  // ... or the row pass found another plan's row table and poisoned its rows
      const int2 stamp = a.row_table[8u * (unsigned)a.slots_per_xcd];
      if ((unsigned)stamp.x != a.plan_tag || (unsigned)stamp.y != ~a.plan_tag) return;
    }
    const int seq = (int)(row / a.num_heads);
    const int ctx = a.context_lens[seq];
    const int planned = a.planned_lens[seq];
    const int tokens_per_part = (int)a.counters[2] * block_size;
    if (planned != ctx) {
      // The plan (which partitions exist) was made from other context lengths than the ones this
      // launch ran on — a workspace reused after context_lens changed.  Not a result: NaN.
      T* bad = static_cast<T*>(a.out) + row * D;
      for (int d = lane; d < D; d += 64) bad[d] = E::from_float(__builtin_nanf(""));
      return;
    }
    nparts = min((ctx + tokens_per_part - 1) / tokens_per_part, a.num_parts);
    if (nparts <= 1) return;  // kModeList wrote this row directly
  }
  const float2* __restrict__ ml = reinterpret_cast<const float2*>(a.ws_ml + 2 * row * a.num_parts);
  const float* __restrict__ wacc = a.ws_acc + row * a.num_parts * D;
  T* op = static_cast<T*>(a.out) + row * D;
  // Every lane reads all the (max, sum) pairs itself (the same addresses in all lanes), so nothing
  // crosses lanes, and no load depends on a condition: indices past the end are clamped to the
  // last partition (and the last column) and what they fetch gets the weight zero — conditional
  // loads made the compiler wait for each before issuing the next.  The first group of accumulator
  // rows is in flight while the maxima arrive: one trip to memory for rows of up to kMergeGroup
  // partitions (a one-sequence launch: 0.6 us less than with the pairs spread over the lanes).
  const int last = nparts - 1;
  float v[kMergeGroup][PER];
  auto load_group = [&](int p0) {
#pragma unroll
    for (int k = 0; k < kMergeGroup; k++) {
      const int64_t pp = min(p0 + k, last);
#pragma unroll
      for (int i = 0; i < PER; i++) v[k][i] = wacc[pp * D + min(lane + 64 * i, D - 1)];
    }
  };
  load_group(0);
  float mg = -FLT_MAX;
  for (int p0 = 0; p0 < nparts; p0 += kMergeGroup) {
    float2 e[kMergeGroup];
#pragma unroll
    for (int k = 0; k < kMergeGroup; k++) e[k] = ml[min(p0 + k, last)];
#pragma unroll
    for (int k = 0; k < kMergeGroup; k++) mg = fmaxf(mg, e[k].x);
  }
  float o[PER];
#pragma unroll
  for (int i = 0; i < PER; i++) o[i] = 0.f;
  float L = 0.f;
  for (int p0 = 0; p0 < nparts; p0 += kMergeGroup) {
    if (p0 > 0) load_group(p0);
    float2 e[kMergeGroup];
#pragma unroll
    for (int k = 0; k < kMergeGroup; k++) e[k] = ml[min(p0 + k, last)];
#pragma unroll
    for (int k = 0; k < kMergeGroup; k++) {
      const float f = p0 + k < nparts ? fast_exp2(e[k].x - mg) : 0.f;
      L = __builtin_fmaf(e[k].y, f, L);
#pragma unroll
      for (int i = 0; i < PER; i++) o[i] = __builtin_fmaf(v[k][i], f, o[i]);
    }
  }
  const float inv = 1.f / (L + 1e-6f);
#pragma unroll
  for (int i = 0; i < PER; i++) {
    const int d = lane + 64 * i;
    if (d < D) op[d] = E::from_float(o[i] * inv);
  }
}

// ---- host side ----------------------------------------------------------------

constexpr int kNumWaves = 4;
constexpr int kBalanceUnit = 4;       // balanced order: consecutive sequences per unit
constexpr double kBalancePerCallBytes = 2e9;
constexpr int kPlainOrderRows = 2048;
// who makes the plan (item list, row table) that sits at the head of the workspace
constexpr int kPlanPerCall = 0;  // this call, for itself
constexpr int kPlanShared = 1;   // this call, for itself and later calls on the same context_lens
constexpr int kPlanReady = 2;    // an earlier call did
constexpr int kListSplitPct = 50;  // kModeList: partitions when the longest row exceeds half the average work per slot

// Waves per workgroup: 4, except head size 64, whose rows stream a quarter of
// the bytes of a head-size-128 row against the same fixed cost and whose 91
// VGPRs leave room for two 8-wave workgroups per CU: measured +1..5 % at batch
// 256 and +37 % at batch 16 (12 heads).  At head sizes 80 / 96 eight waves
// change nothing.  The macros are build-time knobs for A/B runs.
#ifndef PA_DECODE_NW_HEAD64
#define PA_DECODE_NW_HEAD64 8
#endif
#ifndef PA_DECODE_NW_HEAD64_MAXD
#define PA_DECODE_NW_HEAD64_MAXD 64
#endif

static int split_pct() {
  static const int pct = tuning_int("PA_DECODE_SPLIT_PCT", kListSplitPct);  // tuning builds: 0 = always split
  return pct;
}

// workgroups of the decode kernel the GPU holds at a time: 256 CUs x workgroups per CU at the kernel's register count
static int resident_slots(int head_size) {
  return 256 * (head_size <= PA_DECODE_NW_HEAD64_MAXD && PA_DECODE_NW_HEAD64 == 8 ? 2 : 3);
}

// Equal-length rows that are all resident at once need no partitions to end together — unless
// they leave at least half of the CUs idle and each is long: then the partitions are what fills
// the chip.  Measured (32 heads x 128, equal lengths, item-list partitions of 256 tokens against
// one workgroup per row): 3 sequences x 1024 tokens 20.4 -> 17.3 us, 4 x 1024 20.3 -> 17.8 us,
// 4 x 768 17.8 -> 17.0 us; but 4 x 512 13.0 -> 14.4 us, 5 / 6 / 7 x 1024 21.4 -> 22.9 / 22.2 -> 24.6 /
// 23.7 -> 27.9 us, 16 x 1024 at 12 heads x 64 14.0 -> 18.6 us.  Returns the longest context, in
// blocks, from which on the partitions stay (INT_MAX: never).
constexpr int kFewRowsMinElems = 768 * 128;  // tokens x head size
static int few_rows_min_blocks(int64_t rows, int head_size, int block_size) {
  if (rows * 2 > kCUs) return INT_MAX;
  return (kFewRowsMinElems / head_size + block_size - 1) / block_size;
}

template <typename E, int D, int B>
static int launch(const DecodeArgs& a, bool plan_ready, hipStream_t stream) {
  constexpr int NW = D <= PA_DECODE_NW_HEAD64_MAXD ? PA_DECODE_NW_HEAD64 : kNumWaves;
  const unsigned rows = (unsigned)a.num_seqs * (unsigned)a.num_heads;
  // workgroups of the row pass: in the balanced order every XCD gets slots_per_xcd sequences' worth
  const unsigned row_wgs = a.row_table != nullptr ? 8u * (unsigned)a.slots_per_xcd * (unsigned)a.num_heads : rows;
  const bool list = a.num_parts > 1 && a.items != nullptr;
  if ((list || a.row_table != nullptr) && !plan_ready) {
    PlanArgs p;
    p.context_lens = a.context_lens;
    p.num_seqs = a.num_seqs;
    p.block_size = B;
    p.blocks_per_part = a.blocks_per_part;
    p.max_parts = list ? a.num_parts : 1;
    p.split_pct = split_pct();
    p.num_heads = a.num_heads;
    p.resident_slots = resident_slots(D);
    p.keep_parts_min_blocks = few_rows_min_blocks((int64_t)a.num_seqs * a.num_heads, D, B);
    p.items = const_cast<uint32_t*>(a.items);
    p.counters = a.counters;
    p.planned_lens = const_cast<int*>(a.planned_lens);
    p.row_table = const_cast<int2*>(a.row_table);
    p.unit = a.unit;
    p.slots_per_xcd = a.slots_per_xcd;
    p.tag = a.plan_tag;
    int plan_threads = 256;  // a power of two; the balanced order wants one thread per sequence
    while (plan_threads < kPlanThreads && plan_threads < a.num_seqs) plan_threads *= 2;
    hipLaunchKernelGGL(pa_decode_plan_kernel, dim3(1), dim3(plan_threads), 0, stream, p);
    PA_CHECK_LAUNCH("pa_decode_plan_kernel launch");
  }
  if (a.num_parts <= 1) {
    hipLaunchKernelGGL((pa_decode_kernel<E, D, B, NW, kModeRows>), dim3(row_wgs), dim3(NW * 64), 0, stream, a);
    PA_CHECK_LAUNCH("pa_decode_kernel launch");
  } else if (!list) {
    hipLaunchKernelGGL((pa_decode_kernel<E, D, B, NW, kModeParts>), dim3(rows * (unsigned)a.num_parts), dim3(NW * 64),
                       0, stream, a);
    PA_CHECK_LAUNCH("pa_decode_kernel (partitioned) launch");
    hipLaunchKernelGGL((pa_decode_merge_kernel<E, D>), dim3(rows), dim3(64), 0, stream, a, B, 0);
    PA_CHECK_LAUNCH("pa_decode_merge_kernel launch");
  } else {
    hipLaunchKernelGGL((pa_decode_kernel<E, D, B, NW, kModeList>), dim3(a.list_slots + row_wgs), dim3(NW * 64), 0,
                       stream, a);
    PA_CHECK_LAUNCH("pa_decode_kernel (item list) launch");
    hipLaunchKernelGGL((pa_decode_merge_kernel<E, D>), dim3(rows), dim3(64), 0, stream, a, B, 1);
    PA_CHECK_LAUNCH("pa_decode_merge_kernel launch");
  }
  return PA_OK;
}

template <typename E, int D>
static int dispatch_block(int B, const DecodeArgs& a, bool plan_ready, hipStream_t st) {
  switch (B) {
    case 8: return launch<E, D, 8>(a, plan_ready, st);
    case 16: return launch<E, D, 16>(a, plan_ready, st);
    case 32: return launch<E, D, 32>(a, plan_ready, st);
    default: return fail(PA_ERR_UNSUPPORTED, "unsupported block size: %d (supported: 8, 16, 32)", B);
  }
}

template <typename E>
static int dispatch_head(int D, int B, const DecodeArgs& a, bool plan_ready, hipStream_t st) {
  switch (D) {
    case 64: return dispatch_block<E, 64>(B, a, plan_ready, st);
    case 80: return dispatch_block<E, 80>(B, a, plan_ready, st);
    case 96: return dispatch_block<E, 96>(B, a, plan_ready, st);
    case 128: return dispatch_block<E, 128>(B, a, plan_ready, st);
    default: return fail(PA_ERR_UNSUPPORTED, "unsupported head size: %d (supported: 64, 80, 96, 128)", D);
  }
}

// Partition plan.  Splitting pays when the (sequence, head) rows alone cannot
// fill the chip AND each row is a long serial chain of blocks: measured on
// MI355X at 32 rows x 1024 tokens the split pair of kernels takes 10 us
// against 13 us unsplit, at 128 rows there is no gain, and the gain grows with
// the context length.  A partition is never shorter than one block per wave,
// nor than 64 tokens.
// Uniform partitions (few rows).  A row streamed by one CU moves about 40 GB/s however many
// waves work on it (12 waves per row measured the same as 4): the CU's outstanding misses are
// the limit, so a launch of fewer rows than CUs has to spread each row over several CUs.
// A target of 768 workgroups — what the chip holds at a time at head size 128 — beats 1024 (a
// second, mostly empty round) and more.  The partitions are combined by a second launch:
// letting the last partition of a row to finish combine them in the same launch (write-through
// partials, an agent-scope ticket per row) measured 1 us SLOWER on 32 rows, the drain, the
// ticket's round trip and the dependent reads costing more than the launch boundary saved.
// When it pays, measured through the per-call entry point (32 heads x 128, HIP-graph replay, us per launch,
// one workgroup per row -> uniform partitions): 96 rows x 1024 tokens 19.3 -> 14.7, 128 x 1024 20.5 -> 17.6,
// 128 x 768 16.7 -> 14.8, 128 x 512 12.3 -> 11.8; but 160 x 1024 21.1 -> 21.1, 192 x 1024 22.0 -> 22.5,
// 224 x 1024 23.0 -> 26.0, 256 x 512 15.3 -> 17.5, 256 x 1024 25.6 -> 28.8; and 256 x 2048 51.6 -> 48.6.  Up to
// half as many rows as CUs the partitions are what fills the chip; beyond that every CU has a row, and a
// second launch only pays once the rows are long.
constexpr int kSplitTargetWorkgroups = 768;    // aim for this many workgroups
constexpr int kSplitAlwaysRows = 64;           // always for launches of at most this many rows
constexpr int kSplitFewRows = kCUs / 2;        // up to this many rows: when rows x longest context is at least ...
constexpr int kSplitFewRowsMinRowTokens = 64 * 1024;
constexpr int kSplitMaxRows = 256;             // up to this many rows: when it is at least ...
constexpr int kSplitMinRowTokens = 512 * 1024;

struct PartitionPlan {
  int num_parts;
  int blocks_per_part;
  bool list;  // kModeList: partitions of a fixed length, as many as each sequence needs
  // balanced order of the row workgroups (rows and item-list forms); slots_per_xcd = 0: positional order
  int group = 0, unit = 0, slots_per_xcd = 0;
};

// Sequences per work-order group: ~128 workgroups are resident per XCD (32 CUs
// x 4), and group * num_heads of them should span every head.
static int group_for(int num_heads) { return 128 / num_heads < 1 ? 1 : (128 / num_heads > 16 ? 16 : 128 / num_heads); }

// Balanced order: units of up to kBalanceUnit consecutive sequences, for
// batches of at least two units per XCD and at most kMaxBalancedSeqs sequences
// (beyond that the XCDs' shares even out by themselves, and the plan kernel's
// counting would take longer than it saves).
static void plan_balance(PartitionPlan& p, int num_seqs, int num_heads) {
  static const int balance = tuning_int("PA_DECODE_BALANCE", kBalanceUnit);  // tuning builds: 0 = positional order, n = units of n sequences
  if (balance <= 0 || num_seqs > kMaxBalancedSeqs) return;
  int group = group_for(num_heads);
  const int unit = balance < group ? balance : group;
  group = group / unit * unit;
  const int units = (num_seqs + unit - 1) / unit;
  if (units < 16) return;
  const int slots = ((units + 7) / 8 * unit + group - 1) / group * group;
  if (8 * slots > kMaxTableSlots) return;
  p.group = group;
  p.unit = unit;
  p.slots_per_xcd = slots;
}

// kModeList: partitions of a fixed length, as many as each sequence needs, for every batch
// of more than 256 rows whose longest context exceeds one partition.  The fewer rows a launch
// has against the 768 workgroups the chip holds, the more one long row sets the launch time and
// the finer the partitions that pay: at most 16 of at least 256 tokens up to 4096 rows, 8 of
// 512 up to 8192 rows, 4 of 1024 beyond (the partial results take rows x parts x (head size
// + 2) floats of workspace).  Measured (32 heads x 128, against 4 partitions of 1024 tokens
// throughout): 64 sequences of 64 tokens and one of 2048: 56.8 -> 32.6 us; 256 sequences of 256
// and one of 8192: 245 -> 219 us; 32 / 64 sequences of random lengths <= 2048: 96 -> 88 /
// 177 -> 167 us; batches of equal or long-tailed lengths: unchanged within the box's noise.
constexpr int kListPartTokens = 256;   // shortest partition
constexpr int kListMaxParts = 16;      // most partitions per row
constexpr int kListPartsRowBudget = 65536;  // rows x partitions per row stays within this

static int list_max_parts_for(int64_t rows, bool fine) {
  static const int forced = tuning_int("PA_DECODE_LIST_MAX_PARTS", 0);
  if (forced > 0) return forced;
  if (!fine) return 4;
  int parts = kListMaxParts;
  while (parts > 4 && rows * parts > kListPartsRowBudget) parts /= 2;
  return parts;
}

static int waves_for(int head_size) { return head_size <= PA_DECODE_NW_HEAD64_MAXD ? PA_DECODE_NW_HEAD64 : kNumWaves; }

// fine: the caller knows the sum of the context lengths, so the launch can reserve workgroup ids
// for the partitions that sum allows and can tell on the host when none are needed; without it
// finer partitions would cost every equal-length batch thousands of empty ids and the merge
// launch (12 heads x 64, 256 x 1024 tokens: 125 -> 143 us), so the list stays at 4 x 1024 tokens.
static PartitionPlan plan_partitions(int num_seqs, int num_heads, int head_size, int block_size, int max_context_len,
                                     bool fine) {
  const int num_waves = waves_for(head_size);
  const int64_t rows = (int64_t)num_seqs * num_heads;
  const int max_blocks = (max_context_len + block_size - 1) / block_size;
  int min_blocks = num_waves;
  while (min_blocks * block_size < 64) min_blocks *= 2;
  PartitionPlan p = {1, max_blocks > 0 ? max_blocks : 1, false};
  static const int forced = tuning_int("PA_DECODE_FORCE_PARTS", 0);  // tuning builds only
  static const int target_wgs = tuning_int("PA_DECODE_TARGET_WGS", kSplitTargetWorkgroups);
  static const int split_rows = tuning_int("PA_DECODE_SPLIT_ROWS", kSplitMaxRows);
  static const int split_tokens = tuning_int("PA_DECODE_SPLIT_TOKENS", kSplitMinRowTokens);
  static const int few_rows_tokens = tuning_int("PA_DECODE_SPLIT_FEW_ROWS_TOKENS", kSplitFewRowsMinRowTokens);
  const int64_t row_tokens = (int64_t)max_context_len * rows;
  const bool worth_it = rows <= kSplitAlwaysRows || (rows <= kSplitFewRows && row_tokens >= few_rows_tokens) ||
                        (rows <= split_rows && row_tokens >= split_tokens);
  int want = forced > 0 ? forced : (worth_it ? (int)((target_wgs + rows - 1) / rows) : 1);
  static const int list_mode = tuning_int("PA_DECODE_LIST", -1);  // tuning builds: 0 = never use the item list
  const int list_max_parts = list_max_parts_for(rows, fine);
  // 16 partitions of 256 tokens, 8 of 512, 4 of 1024: a row of up to 4096 tokens uses them all
  static const int forced_part_tokens = tuning_int("PA_DECODE_LIST_PART_TOKENS", 0);
  const int list_part_tokens = forced_part_tokens > 0 ? forced_part_tokens : kListPartTokens * kListMaxParts / list_max_parts;
  int list_bpp = std::max(list_part_tokens / block_size, (max_blocks + list_max_parts - 1) / list_max_parts);
  list_bpp = (list_bpp + num_waves - 1) / num_waves * num_waves;
  // some sequence can have a second partition; with thousands of rows (ten and more per resident
  // workgroup) a row of up to 1024 tokens cannot hold up the launch, whatever the others are
  const bool long_contexts = max_blocks > list_bpp && (rows <= 4096 || max_context_len > 1024);
  if (want <= 1 && list_mode != 0 && long_contexts && num_seqs < (1 << (32 - kItemPartBits))) {
    const int parts = (max_blocks + list_bpp - 1) / list_bpp;
    if (parts > 1 && rows * parts < (1ll << 31)) {
      PartitionPlan lp = {parts, list_bpp, true};
      plan_balance(lp, num_seqs, num_heads);
      return lp;
    }
  }
  if (want <= 1) plan_balance(p, num_seqs, num_heads);  // one workgroup per row: a big batch, or short contexts
  if (want <= 1 || max_blocks <= min_blocks) return p;
  int bpp = (max_blocks + want - 1) / want;
  if (bpp < min_blocks) bpp = min_blocks;
  bpp = (bpp + num_waves - 1) / num_waves * num_waves;  // whole rounds of the waves
  p.blocks_per_part = bpp;
  p.num_parts = (max_blocks + bpp - 1) / bpp;
  if (p.num_parts <= 1) p = PartitionPlan{1, max_blocks, false};
  return p;
}

// Workspace: the balanced order's row table [8][slots_per_xcd] of int2 first
// (if any); with partitions, accumulators [rows * parts][D], then (max, sum)
// pairs [rows * parts][2], fp32; kModeList appends the context lengths the plan
// was made from [num_seqs], the item list [num_seqs * parts] and the three counters.
static size_t workspace_bytes_for(int64_t rows, const PartitionPlan& p, int head_size, int64_t num_seqs) {
  size_t n = p.slots_per_xcd > 0 ? ((size_t)8 * p.slots_per_xcd + 1) * sizeof(int2) : 0;  // + the stamp
  if (p.num_parts > 1) n += (size_t)rows * p.num_parts * (head_size + 2) * sizeof(float);
  if (p.num_parts > 1 && p.list) n += (size_t)num_seqs * (p.num_parts + 1) * sizeof(uint32_t) + 3 * sizeof(unsigned);
  return n;
}

// enough for either plan: with and without the caller's total context length
static size_t workspace_bytes_any(int num_seqs, int num_heads, int head_size, int block_size, int max_context_len) {
  size_t n = 0;
  for (bool fine : {false, true}) {
    const PartitionPlan p = plan_partitions(num_seqs, num_heads, head_size, block_size, max_context_len, fine);
    n = std::max(n, workspace_bytes_for((int64_t)num_seqs * num_heads, p, head_size, num_seqs));
  }
  return n;
}

// Equal long contexts on a batch between one and two residencies of rows (the caller's total says so:
// every sequence within 3 % of the longest) do better as two uniform partitions per row than as the item
// list's partitions of 1024 tokens: no plan kernel, half the partial results, a merge of two.  Measured
// (32 heads x 128, fp16, the shared-plan entry): 32 x 4096 tokens 359 -> 341 us, 40 x 4096 448 -> 422,
// 32 x 2048 179 -> 173, 32 x 8192 705 -> 672; no gain at 24 or 48 sequences (exactly one, two residencies),
// a loss of 2 % from 64 sequences on, and 9 % on ragged lengths, which keep the list.
static bool equal_long_contexts_take_two_parts(int64_t rows, int64_t longest_blocks, int64_t total_blocks, int num_seqs,
                                               int head_size) {
  static const int enabled = tuning_int("PA_DECODE_EQUAL_LONG_TWO_PARTS", 1);
  const int64_t resident = resident_slots(head_size);
  return enabled != 0 && rows > resident && rows < 2 * resident && longest_blocks * num_seqs * 100 <= total_blocks * 103;
}

// The step's new token of every sequence, for the append form: [num_seqs][num_heads][head_size]
// rows with these strides in elements.
struct NewTokens {
  const void* key;
  const void* value;
  int64_t key_stride;
  int64_t value_stride;
};

static int decode_entry(void* out, const void* query, const void* key_cache, const void* value_cache, float scale,
                        const int32_t* block_tables, const int32_t* context_lens, int num_seqs, int num_heads,
                        int head_size, int block_size, int max_num_blocks_per_seq, int max_context_len,
                        int64_t q_stride, int dtype, void* workspace, size_t workspace_bytes, int plan_mode,
                        int64_t total_context_len, void* stream, const NewTokens* new_tokens = nullptr) {
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
  const int64_t rows = (int64_t)num_seqs * num_heads;
  PA_REQUIRE(rows < (1ll << 31), "num_seqs * num_heads too large");
  const int es = elem_size(dtype);

  DecodeArgs a;
  a.out = out;
  a.q = query;
  a.k_cache = key_cache;
  a.v_cache = value_cache;
  a.block_tables = block_tables;
  a.context_lens = context_lens;
  a.ws_acc = nullptr;
  a.ws_ml = nullptr;
  a.items = nullptr;
  a.counters = nullptr;
  a.planned_lens = nullptr;
  a.list_slots = 0;
  a.q_stride = q_stride;
  a.scale_log2e = scale * 1.4426950408889634f;
  a.num_seqs = num_seqs;
  a.num_heads = num_heads;
  a.max_blocks_per_seq = max_num_blocks_per_seq;
  a.q_vec_ok = aligned16(query) && ((q_stride * (int64_t)es) % 16 == 0) && ((head_size * es) % 16 == 0);
  a.num_parts = 1;
  a.blocks_per_part = max_num_blocks_per_seq;
  a.group = group_for(num_heads);
  a.row_table = nullptr;
  a.slots_per_xcd = 0;
  a.unit = 1;
  static const int plain_rows = tuning_int("PA_DECODE_PLAIN_ROWS", kPlainOrderRows);  // tuning builds only
  a.plain_order = rows <= plain_rows ? 1 : 0;
  a.plan_tag = 0;
  a.new_k = nullptr;
  a.new_v = nullptr;
  a.new_k_stride = a.new_v_stride = 0;
  a.new_vec_ok = 0;
  if (new_tokens != nullptr) {
    PA_REQUIRE(new_tokens->key && new_tokens->value, "null key / value argument");
    PA_REQUIRE(new_tokens->key_stride >= (int64_t)num_heads * head_size && new_tokens->value_stride >= (int64_t)num_heads * head_size,
               "key_stride (%lld) / value_stride (%lld) shorter than a token row of %d x %d elements",
               (long long)new_tokens->key_stride, (long long)new_tokens->value_stride, num_heads, head_size);
    PA_REQUIRE((reinterpret_cast<uintptr_t>(new_tokens->key) % es) == 0 && (reinterpret_cast<uintptr_t>(new_tokens->value) % es) == 0,
               "key / value must be aligned to the element size");
    a.new_k = new_tokens->key;
    a.new_v = new_tokens->value;
    a.new_k_stride = new_tokens->key_stride;
    a.new_v_stride = new_tokens->value_stride;
    a.new_vec_ok = aligned16(new_tokens->key) && ((new_tokens->key_stride * (int64_t)es) % 16 == 0) && ((head_size * es) % 16 == 0);
  }
  if (workspace != nullptr) {
    PA_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 3u) == 0, "workspace must be 4-byte aligned");
    PartitionPlan p = plan_partitions(num_seqs, num_heads, head_size, block_size, max_context_len,
                                      total_context_len >= 0);
    if (p.list && total_context_len >= 0) {
      // The plan kernel's rule, applied here with what the caller knows: no sequence is
      // longer than max_context_len, and the batch holds at least total / block_size blocks.
      // If even so the longest row cannot hold up the launch, the item list, its workgroup
      // slots and the merge launch are left out altogether.
      const int64_t longest = (max_context_len + block_size - 1) / block_size;
      const int64_t blocks = total_context_len / block_size;
      const bool no_straggler = rows <= resident_slots(head_size) && longest < few_rows_min_blocks(rows, head_size, block_size) &&
                                longest * num_seqs * 100 <= blocks * 125;
      if (no_straggler || longest * 100 * resident_slots(head_size) <= blocks * num_heads * split_pct()) {
        p.num_parts = 1;
        p.list = false;
        p.blocks_per_part = longest > 0 ? (int)longest : 1;
      } else if (equal_long_contexts_take_two_parts(rows, longest, blocks, num_seqs, head_size)) {
        const int nw = waves_for(head_size);
        const int bpp = (int)((longest + 1) / 2 + nw - 1) / nw * nw;  // whole rounds of the waves
        p = PartitionPlan{(int)((longest + bpp - 1) / bpp), bpp, false};
      }
    }
    if ((reinterpret_cast<uintptr_t>(workspace) & 7u) != 0) p.slots_per_xcd = 0;  // the table holds 8-byte entries
    if (rows * p.num_parts >= (1ll << 31)) p = PartitionPlan{1, p.blocks_per_part, false};
    if (workspace_bytes_for(rows, p, head_size, num_seqs) > workspace_bytes) p.slots_per_xcd = 0;  // retry without the table
    if ((p.num_parts > 1 || p.slots_per_xcd > 0) && workspace_bytes_for(rows, p, head_size, num_seqs) <= workspace_bytes) {
      unsigned tag = 0x9e3779b9u;  // the sizes a plan depends on, mixed into one word
      for (unsigned v : {(unsigned)num_seqs, (unsigned)num_heads, (unsigned)block_size, (unsigned)p.num_parts,
                         (unsigned)p.blocks_per_part, (unsigned)p.list, (unsigned)p.unit, (unsigned)p.slots_per_xcd})
        tag = (tag ^ v) * 0x01000193u + (tag >> 15);
      a.plan_tag = tag | 1u;
      char* base = static_cast<char*>(workspace);
      if (p.slots_per_xcd > 0) {
        // A plan made for this call alone has to earn its few microseconds: only for
        // launches that stream a couple of gigabytes.  A plan shared by the layers of
        // a step always pays.
        const double streamed = 2.0 * (double)rows * max_context_len * head_size * es;
        if (plan_mode != kPlanPerCall || streamed >= kBalancePerCallBytes) {
          a.row_table = reinterpret_cast<const int2*>(base);
          a.slots_per_xcd = p.slots_per_xcd;
          a.unit = p.unit;
          a.group = p.group;
        }
        base += ((size_t)8 * p.slots_per_xcd + 1) * sizeof(int2);
      }
      if (p.num_parts > 1) {
        a.num_parts = p.num_parts;
        a.blocks_per_part = p.blocks_per_part;
        a.ws_acc = reinterpret_cast<float*>(base);
        a.ws_ml = a.ws_acc + (size_t)rows * p.num_parts * head_size;
        if (p.list) {
          uint32_t* tail = reinterpret_cast<uint32_t*>(a.ws_ml + (size_t)rows * p.num_parts * 2);
          a.planned_lens = reinterpret_cast<const int*>(tail);
          tail += num_seqs;
          a.items = tail;
          a.counters = tail + (size_t)num_seqs * p.num_parts;
          // One workgroup id per (sequence, partition >= 1, head) that could exist.  A sequence of
          // n blocks has at most n / blocks_per_part partitions beyond its first, so with the sum
          // of the context lengths known the batch has at most (total / block_size + num_seqs) /
          // blocks_per_part of them: far fewer ids to dispatch than num_seqs * (parts - 1).
          int64_t pairs = (int64_t)num_seqs * (p.num_parts - 1);
          if (total_context_len >= 0)
            pairs = std::min<int64_t>(pairs, (total_context_len / block_size + num_seqs) / p.blocks_per_part);
          a.list_slots = (unsigned)(pairs * num_heads);
        }
      }
    }
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case PA_DTYPE_F32: return dispatch_head<ElemF32>(head_size, block_size, a, plan_mode == kPlanReady, st);
    case PA_DTYPE_F16: return dispatch_head<ElemF16>(head_size, block_size, a, plan_mode == kPlanReady, st);
    default: return dispatch_head<ElemBF16>(head_size, block_size, a, plan_mode == kPlanReady, st);
  }
}

// ---- the library's own scratch, for callers that cannot allocate ---------------
//
// pa_single_query_cached_kv_attention_static runs the _ws form on a scratch
// buffer that lives in the library's code object: every GPU the library is
// loaded on holds its own kStaticScratchBytes copy of `scratch` until the
// process ends.  In host code the array's name is only a handle for
// hipGetSymbolAddress; the kernels are given the device address that call
// returns for the current device, kept per device in static_scratch_ptr.
constexpr size_t kStaticScratchBytes = PA_DECODE_STATIC_SCRATCH_BYTES;
constexpr int kMaxDevices = 64;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
static std::atomic<void*> static_scratch_ptr[kMaxDevices];

// Device address of the current device's copy of `scratch` (looked up once per device).
static int static_scratch(void** ptr) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail(PA_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e));
  if (dev < 0 || dev >= kMaxDevices) return fail(PA_ERR_UNSUPPORTED, "device index %d beyond the %d this library keeps scratch for", dev, kMaxDevices);
  void* p = static_scratch_ptr[dev].load(std::memory_order_acquire);
  if (p == nullptr) {
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr)
      return fail(PA_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
    static_scratch_ptr[dev].store(p, std::memory_order_release);  // racing threads store the same address
  }
  *ptr = p;
  return PA_OK;
}

}  // namespace pa

extern "C" int pa_single_query_cached_kv_attention_static(
    void* out, const void* query, const void* key_cache, const void* value_cache, float scale,
    const int32_t* block_tables, const int32_t* context_lens, int num_seqs, int num_heads, int head_size,
    int block_size, int max_num_blocks_per_seq, int max_context_len, int64_t q_stride, int dtype, void* stream) {
  pa::clear_error();
  size_t need = 0;
  if (num_seqs > 0 && num_heads > 0 && max_context_len > 0 && (head_size == 64 || head_size == 80 || head_size == 96 || head_size == 128) &&
      (block_size == 8 || block_size == 16 || block_size == 32)) {
    need = pa::workspace_bytes_any(num_seqs, num_heads, head_size, block_size, max_context_len);
  }
  if (need > pa::kStaticScratchBytes)
    return pa::fail(PA_ERR_INVALID_ARGUMENT,
                    "this launch needs %zu bytes of scratch, the library's static scratch holds %zu "
                    "(num_seqs %d x num_heads %d, head size %d, max_context_len %d): use "
                    "pa_single_query_cached_kv_attention_ws with a caller workspace",
                    need, pa::kStaticScratchBytes, num_seqs, num_heads, head_size, max_context_len);
  void* ws = nullptr;
  if (need > 0) {
    const int rc = pa::static_scratch(&ws);
    if (rc != PA_OK) return rc;
  }
  return pa::decode_entry(out, query, key_cache, value_cache, scale, block_tables, context_lens, num_seqs, num_heads,
                          head_size, block_size, max_num_blocks_per_seq, max_context_len, q_stride, dtype, ws,
                          ws != nullptr ? pa::kStaticScratchBytes : 0, pa::kPlanPerCall, -1, stream);
}

extern "C" int pa_single_query_cached_kv_attention(
    void* out, const void* query, const void* key_cache, const void* value_cache, float scale,
    const int32_t* block_tables, const int32_t* context_lens, int num_seqs, int num_heads, int head_size,
    int block_size, int max_num_blocks_per_seq, int max_context_len, int64_t q_stride, int dtype, void* stream) {
  return pa::decode_entry(out, query, key_cache, value_cache, scale, block_tables, context_lens, num_seqs, num_heads,
                          head_size, block_size, max_num_blocks_per_seq, max_context_len, q_stride, dtype, nullptr, 0,
                          pa::kPlanPerCall, -1, stream);
}

extern "C" size_t pa_decode_workspace_bytes(int num_seqs, int num_heads, int head_size, int block_size,
                                            int max_context_len) {
  if (num_seqs <= 0 || num_heads <= 0 || head_size <= 0 || block_size <= 0 || max_context_len <= 0) return 0;
  return pa::workspace_bytes_any(num_seqs, num_heads, head_size, block_size, max_context_len);
}

extern "C" int pa_single_query_cached_kv_attention_ws(
    void* out, const void* query, const void* key_cache, const void* value_cache, float scale,
    const int32_t* block_tables, const int32_t* context_lens, int num_seqs, int num_heads, int head_size,
    int block_size, int max_num_blocks_per_seq, int max_context_len, int64_t q_stride, int dtype, void* workspace,
    size_t workspace_bytes, void* stream) {
  return pa::decode_entry(out, query, key_cache, value_cache, scale, block_tables, context_lens, num_seqs, num_heads,
                          head_size, block_size, max_num_blocks_per_seq, max_context_len, q_stride, dtype, workspace,
                          workspace_bytes, pa::kPlanPerCall, -1, stream);
}

extern "C" int pa_single_query_cached_kv_attention_planned(
    void* out, const void* query, const void* key_cache, const void* value_cache, float scale,
    const int32_t* block_tables, const int32_t* context_lens, int num_seqs, int num_heads, int head_size,
    int block_size, int max_num_blocks_per_seq, int max_context_len, int64_t q_stride, int dtype, void* workspace,
    size_t workspace_bytes, int plan_ready, int64_t total_context_len, void* stream) {
  return pa::decode_entry(out, query, key_cache, value_cache, scale, block_tables, context_lens, num_seqs, num_heads,
                          head_size, block_size, max_num_blocks_per_seq, max_context_len, q_stride, dtype, workspace,
                          workspace_bytes, plan_ready ? pa::kPlanReady : pa::kPlanShared, total_context_len, stream);
}

extern "C" int pa_append_single_query_cached_kv_attention(
    void* out, const void* query, const void* key, const void* value, void* key_cache, void* value_cache, float scale,
    const int32_t* block_tables, const int32_t* context_lens, int num_seqs, int num_heads, int head_size,
    int block_size, int max_num_blocks_per_seq, int max_context_len, int64_t q_stride, int64_t key_stride,
    int64_t value_stride, int dtype, void* workspace, size_t workspace_bytes, int plan_ready,
    int64_t total_context_len, void* stream) {
  const pa::NewTokens nt = {key, value, key_stride, value_stride};
  return pa::decode_entry(out, query, key_cache, value_cache, scale, block_tables, context_lens, num_seqs, num_heads,
                          head_size, block_size, max_num_blocks_per_seq, max_context_len, q_stride, dtype, workspace,
                          workspace != nullptr ? workspace_bytes : 0,
                          plan_ready > 0 ? pa::kPlanReady : (plan_ready < 0 ? pa::kPlanPerCall : pa::kPlanShared),
                          total_context_len, stream, &nt);
}
