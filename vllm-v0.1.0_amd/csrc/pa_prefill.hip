// Prompt attention for gfx950: causal self-attention inside each prompt of a
// flat token batch, on the matrix cores.
//
// Replaces what PagedAttention.multi_query_kv_attention hands to xformers'
// memory_efficient_attention_forward with a BlockDiagonalCausalMask
// (vllm/model_executor/layers/attention.py:44-70): query / key / value are the
// [num_prompt_tokens, H, D] slices of the fused qkv projection (free row
// strides), the output is dense.
//
// One workgroup = 4 or 8 waves = 128 or 256 query rows of one (prompt, head);
// each wave owns 32 rows.  Key/value tiles of 64 rows (32 for fp32) go global -> VGPR ->
// LDS, double buffered, the loads of tile t+1 in flight while tile t is
// consumed.  Per tile and wave:
//   S^T = K . Q^T   v_mfma_f32_32x32x16_{f16,bf16}: the KEY index lands on the
//                   accumulator registers and the QUERY row on the lane, so a
//                   row's softmax statistics are lane-local (one exchange
//                   between lanes l and l+32 for the maximum);
//   online softmax  in the log2 domain, fp32;
//   O^T += V^T . P^T  the S^T accumulator, converted to 16 bit, is the B
//                   operand as it stands (its register index is the summation
//                   index); V^T fragments come from the row-major V image by
//                   ds_read_b64_tr_b16.
// fp32 tensors take the same skeleton with v_mfma_f32_32x32x2_f32 (exact fp32
// products and sums) and element reads of the LDS images.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "pa_common.h"

namespace {

using namespace pa;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int kMaxSeqs = 256;   // prompts per launch (table rides in the kernel arguments)
constexpr int kMaxWork = 512;   // (prompt, query block) items per launch
constexpr int kChipCUs = 256;

struct PromptArgs {
  void* out;
  const void* q;
  const void* k;
  const void* v;
  int64_t q_stride, k_stride, v_stride;  // elements between token rows
  float c;                               // |scale| * log2(e)
  int q_mode;                            // 0: q as stored, 1: -q (scale < 0), 2: q = 0 (scale == 0)
  int num_heads;
  int num_units;                         // (work item, head) pairs of this launch
  int32_t cu[kMaxSeqs + 1];              // first token of each prompt of this launch
  uint32_t work[kMaxWork];               // (prompt << 16) | query block, heaviest first
};
static_assert(sizeof(PromptArgs) <= 4096, "kernel arguments are limited to 4 KiB");

struct OpsF16 {
  static constexpr int ES = 2;
  static constexpr uint32_t kOnes = 0x3C003C00u;  // two fp16 ones
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    f16x2 h = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, h);
  }
};
struct OpsBF16 {
  static constexpr int ES = 2;
  static constexpr uint32_t kOnes = 0x3F803F80u;  // two bf16 ones
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    bf16x2 h = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, h);
  }
};
struct OpsF32 {
  static constexpr int ES = 4;
};

// Value of lane l ^ 32 (v_permlane32_swap: no LDS round trip).
__device__ __forceinline__ float other_half(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // {x[l % 32], x[l % 32 + 32]} in every lane
  const float lo = __builtin_bit_cast(float, (uint32_t)sw[0]), hi = __builtin_bit_cast(float, (uint32_t)sw[1]);
  return (threadIdx.x & 32) ? lo : hi;
}

// How far (log2 units) a row's true maximum may run ahead of the one its
// weights are scaled by before the accumulators are rescaled.
constexpr float kDeferLog2 = 5.f;

#ifndef PA_PROMPT_SCHED
#define PA_PROMPT_SCHED 1
#endif
#ifndef PA_PROMPT_PREFETCH
#define PA_PROMPT_PREFETCH 4
#endif
#ifndef PA_PROMPT_CLUSTER_PRIO
#define PA_PROMPT_CLUSTER_PRIO 1
#endif
#ifndef PA_PROMPT_PREFETCH_TIGHT
#define PA_PROMPT_PREFETCH_TIGHT 3
#endif
// K fragments in flight ahead of their MFMA: PA_PROMPT_PREFETCH, or _TIGHT where the register
// budget is cut for more than two waves per SIMD

#ifndef PA_PROMPT_PERSISTENT
#define PA_PROMPT_PERSISTENT 0
#endif
// 1 (variant builds): where a workgroup has its CU to itself (8 waves, 16 bit, head size above 64) the
// launch is one workgroup per CU, each working through several (work item, head) units; the K/V tiles of
// all its units form one stream, so the first tiles of the next unit are staged under the last tiles of
// the current one and the next queries are fetched under the current unit's epilogue.  Measured against
// one workgroup per unit (bf16, 32 heads x 128): 32 x 512 +9 %, 16 x 1024 +1.6 %, 8 x 2048 +0.7 %,
// 4 x 4096 -1.2 %, one 16k prompt 0: a unit switch still costs ~8.7 k cycles per wave (the wait for the
// queries is the larger part), against ~12 k for a workgroup's start and end, and all eight waves of the
// CU spend them together.  Not the shipped form.

// -DPA_PROMPT_TIMING (variant builds only): every wave adds the shader cycles it spent in each phase of
// its tile loop to prompt_phase_cycles; pa_prompt_timing_read() returns and clears the totals.
// Phases: 0 issue of the next tile's global loads, 1 scores, 2 mask + softmax, 3 V^T P^T, 4 commit to LDS
// (waits for the global loads), 5 barrier, 6 wave-tiles counted, 7 prologue, 8 epilogue, 9 waves counted.
#ifdef PA_PROMPT_TIMING
constexpr int kTimingWaves = 32768;  // waves of one launch that find a slot of their own (the rest are not recorded)
__device__ unsigned long long prompt_phase_cycles[kTimingWaves][10];
#define PA_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PA_T_ADD(slot, t1, t0) phase_acc[slot] += (t1) - (t0)
#else
#define PA_T(var)
#define PA_T_ADD(slot, t1, t0)
#endif

template <typename X, int D, int NW>
struct PromptGeometry {
  static constexpr int THREADS = 64 * NW;
  static constexpr int QBLOCK = 32 * NW;     // query rows per workgroup
  static constexpr int ES = X::ES;
  static constexpr bool F32 = ES == 4;
  static constexpr int KT = F32 ? 32 : 64;   // keys per tile
  static constexpr int KB = KT / 32;         // 32-key blocks per tile
  static constexpr int DB = (D + 31) / 32;   // 32-row blocks of O^T
  static constexpr int DP = DB * 32;
  // K image: rows read by ds_read_b128 (16 bit: stride = odd multiple of 16 B)
  // or ds_read_b32 (fp32: odd number of dwords) by 32 lanes on 32 rows.
  static constexpr int KSTR = F32 ? D * 4 + 4 : D * 2 + 16;
  // V image: 16 bit rows are read transposed, 4 rows x 64 B per 32-lane half;
  // a stride of 64 or 192 (mod 256) bytes spreads them over all banks.
  static constexpr int VSTR = F32 ? DP * 4 : ((DP * 2) % 128 == 0 ? DP * 2 + 64 : DP * 2);
  static constexpr int KBYTES = KT * KSTR;
  static constexpr int VBYTES = KT * VSTR;
  static constexpr int BUF = KBYTES + VBYTES;
  // Epilogue of the 16-bit kernels: a wave's 32 output rows pass through a strip of LDS of its
  // own (row stride OSTR: 16-byte writes by 8 lanes on 8 rows spread over all banks) so that
  // they leave as whole rows.
  static constexpr int OSTR = D * 2 + 16;
  static constexpr int OUT_BYTES = F32 ? 0 : NW * 32 * OSTR;
  // Several units per workgroup: the strips cannot share memory with the images, which already hold
  // the next unit's first tile when a unit's rows leave.
  static constexpr bool PERSIST = PA_PROMPT_PERSISTENT && !F32 && NW == 8 && D != 64;
  static constexpr int STRIP_BASE = PERSIST ? 2 * BUF : 0;
  static constexpr int LDS_BYTES = PERSIST ? 2 * BUF + OUT_BYTES : (2 * BUF > OUT_BYTES ? 2 * BUF : OUT_BYTES);
  static_assert(LDS_BYTES <= 160 * 1024, "LDS of a CU");
  static constexpr int CPR = D * ES / 16;    // 16-byte chunks per row
  static constexpr int TOTAL = KT * CPR;
  static constexpr int NCH = (TOTAL + THREADS - 1) / THREADS;
  // rows a thread's consecutive chunks are apart, when that is the same for every thread
  static constexpr bool UNIFORM_ROWS = THREADS % CPR == 0;
  static constexpr int ROWS_PER_PASS = THREADS / CPR;
  static constexpr int NQ = F32 ? D / 2 : D / 4;  // dwords of q per lane
  // Waves per SIMD the register budget is cut for.  At head size 64 the 16-bit kernel fits 128
  // registers, and two 8-wave workgroups then share a CU (their LDS images are 42 KiB each).
  // An 8-wave workgroup's two waves per SIMD compete for issue slots: a wave inside one of its
  // MFMA clusters (S^T, O^T) runs at priority 1, so the partner's softmax fills the gaps
  // between its MFMAs instead of delaying them (+1 % on long prompts; nothing at 4 waves).
  static constexpr bool CLUSTER_PRIO = PA_PROMPT_CLUSTER_PRIO && NW == 8 && !F32;
  static constexpr int MIN_WAVES = (!F32 && D == 64 && NW == 8) ? 4 : 2;
  // Tiles whose global loads are in flight ahead of the tile being computed: two at head size 128 with 8 waves
  // (a second set of staging registers: 245 in all, no spill), one elsewhere.  Measured, bf16, 32 heads: 16k prompt
  // +2.8 %, 8 x 2048 +2.7 %, 4 x 4096 +1.1 %, short prompts unchanged; head size 96 mixed (16k +1.4 %, 8 x 2048 -3 %),
  // so it stays at one there.  A per-phase cycle count of the loop (tools/prompt_phase_cycles.py) shows why the gain
  // is small: the ~500 cycles a wave spends in commit() per tile are not the wait for its loads — they stay with
  // two tiles in flight — but the tile's 32 KiB going through the CU's 16-byte LDS store path (~79 bytes per
  // clock) with all eight waves storing at once.  Committing earlier, under the softmax phase, only moved that
  // time in front of the next LDS reads (16k prompt -1.4 %, 4 x 4096 -4 %, head size 96 -5 ... -8 %): not kept.
#ifndef PA_PROMPT_TILES_AHEAD
#define PA_PROMPT_TILES_AHEAD 2
#endif
  static constexpr int TILES_AHEAD = (PA_PROMPT_TILES_AHEAD == 2 && !F32 && D == 128 && NW == 8) ? 2 : 1;
  static_assert(KBYTES % 16 == 0 && VSTR % 16 == 0, "images must keep 16-byte alignment");
};

template <typename X, int D, int NW>
__global__ __launch_bounds__(64 * NW, (PromptGeometry<X, D, NW>::MIN_WAVES)) void pa_prompt_kernel(const PromptArgs a) {
  using G = PromptGeometry<X, D, NW>;
  constexpr int THREADS = G::THREADS, kQBlock = G::QBLOCK;
  constexpr int ES = G::ES;
  constexpr bool F32 = G::F32;
  constexpr int KT = G::KT, KB = G::KB, DB = G::DB, DP = G::DP;
  constexpr int KSTR = G::KSTR, VSTR = G::VSTR, KBYTES = G::KBYTES, BUF = G::BUF;
  constexpr int CPR = G::CPR, TOTAL = G::TOTAL, NCH = G::NCH, NQ = G::NQ;
  extern __shared__ __align__(16) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: what depends on the wave only stays out of the vector registers
  const int r = lane & 31, hh = lane >> 5;
#ifdef PA_PROMPT_TIMING
  unsigned long long phase_acc[10] = {};
#endif
  PA_T(t_start);
  // A unit = one (work item, head) pair: kQBlock query rows of one head of one prompt.  Unit u of the
  // launch is work item u / num_heads, head u % num_heads (heaviest items first).  A persistent
  // workgroup takes one unit of every round of gridDim.x units, from alternating ends, which evens
  // out the sums; otherwise the workgroup is its one unit.  Everything here is workgroup-uniform.
  struct Unit {
    int valid, head, n, n_tiles, qb;
    int64_t t0;
  };
  auto load_unit = [&](int j) {
    int u;
    if constexpr (G::PERSIST) {
      u = j * (int)gridDim.x + ((j & 1) ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x);
    } else {
      u = j == 0 ? (int)blockIdx.x : a.num_units;
    }
    Unit it;
    it.valid = u < a.num_units;
    if (!it.valid) u = 0;  // table reads stay in range
    const unsigned wi = (unsigned)u / (unsigned)a.num_heads;
    it.head = (int)((unsigned)u - wi * (unsigned)a.num_heads);
    const uint32_t item = a.work[wi];
    const int seq = (int)(item >> 16);
    it.qb = (int)(item & 0xffffu);
    it.t0 = a.cu[seq];
    it.n = a.cu[seq + 1] - a.cu[seq];
    it.n_tiles = (min(it.n, it.qb * kQBlock + kQBlock) + KT - 1) / KT;
    return it;
  };
  // a place in the workgroup's stream of K/V tiles: tile kt of its unit number j
  struct Cursor {
    Unit u;
    int j, kt;
    const unsigned char* kp;  // first key / value row of the tile, this head's columns: carried from tile to
    const unsigned char* vp;  // tile by one scalar addition each, so that issuing a tile costs no multiplications
  };
  const int64_t k_row_bytes = a.k_stride * ES, v_row_bytes = a.v_stride * ES;  // below 16 MiB (checked by the host)
  auto first_tile = [&](Cursor& c) {
    c.kt = 0;
    c.kp = (const unsigned char*)a.k + c.u.t0 * k_row_bytes + (int64_t)c.u.head * D * ES;
    c.vp = (const unsigned char*)a.v + c.u.t0 * v_row_bytes + (int64_t)c.u.head * D * ES;
  };
  auto advance = [&](Cursor& c) {
    if (++c.kt == c.u.n_tiles) {
      c.u = load_unit(++c.j);
      first_tile(c);
    } else {
      c.kp += KT * k_row_bytes;
      c.vp += KT * v_row_bytes;
    }
  };

  // Columns D..DP-1 of the V image (head size 80) are never staged: clear
  // both buffers once so the padded rows of O^T stay finite.
  if constexpr (DP != D) {
    for (int i = tid * 16; i < G::LDS_BYTES; i += THREADS * 16) *(u32x4*)(smem + i) = u32x4{0, 0, 0, 0};
    __syncthreads();
  }

  // ---- query fragments (registers, whole kernel) ---------------------------
  // 16 bit: dwords 4ks..4ks+3 = Q[q0 + r][16 ks + 8 hh .. +7], the B operand of
  // k-step ks.  fp32: dword ks = Q[q0 + r][hh D/2 + ks] (the summation order
  // over d is a free choice as long as K uses the same).
  uint32_t qf[NQ];
  // Inside the tile loop (several units per workgroup) the lane-dependent addresses of the query fetch and of
  // the epilogue must not be hoisted out of it: that would hold a dozen registers over the whole loop.
  auto opaque_lane = [&]() {
    int ln = lane;
    if constexpr (G::PERSIST) asm volatile("" : "+v"(ln));
    return ln;
  };
  auto load_q = [&](const Unit& it) {
    const int ln = opaque_lane();
    const int r = ln & 31, hh = ln >> 5;
    const int q0 = it.qb * kQBlock + wave * 32;
    const bool ok = q0 + r < it.n && a.q_mode != 2;
    const unsigned char* qrow =
        (const unsigned char*)a.q + ((it.t0 + q0 + r) * a.q_stride + (int64_t)it.head * D) * ES;
#pragma unroll
    for (int i = 0; i < NQ / 4; ++i) {
      const int byte_off = F32 ? (hh * (D / 2) + 4 * i) * 4 : (16 * i + 8 * hh) * 2;
      u32x4 v = {0, 0, 0, 0};
      if (ok) v = *(const u32x4*)(qrow + byte_off);
      qf[4 * i + 0] = v.x;
      qf[4 * i + 1] = v.y;
      qf[4 * i + 2] = v.z;
      qf[4 * i + 3] = v.w;
    }
  };
  // A negative scale is applied as -q (the kernel's factor is |scale|).  Kept apart from the fetch, behind a
  // uniform branch: nothing else touches the fragments before the first MFMA, so a fetch issued before a
  // unit's epilogue is waited for after it.
  auto negate_q_if_asked = [&]() {
    if (a.q_mode == 1) {
      const uint32_t flip = F32 ? 0x80000000u : 0x80008000u;
#pragma unroll
      for (int i = 0; i < NQ; ++i) qf[i] ^= flip;
    }
  };

  // ---- K/V staging: global -> registers -> LDS ------------------------------
  constexpr int AHEAD = G::TILES_AHEAD;
  u32x4 kreg[NCH], vreg[NCH];
  [[maybe_unused]] u32x4 kreg2[AHEAD == 2 ? NCH : 1], vreg2[AHEAD == 2 ? NCH : 1];  // the second tile in flight
  // Loads never branch on a lane: in the prompt's last tile, rows past its end
  // (and the idle lanes of a partial chunk pass) re-read the last row, and
  // commit() replaces what they fetched by zeros.  Everywhere else a thread's
  // chunks sit at wave-uniform distances, so one 32-bit lane offset serves all
  // of them and the rest of the address arithmetic is scalar.
  const uint32_t lane_off_k = (uint32_t)((tid / CPR) * a.k_stride * ES + (tid % CPR) * 16);
  const uint32_t lane_off_v = (uint32_t)((tid / CPR) * a.v_stride * ES + (tid % CPR) * 16);
  auto issue = [&](auto& kreg, auto& vreg, const Cursor& c) {
    const int rows_left = c.u.n - c.kt * KT;  // rows of the prompt from this tile's first on (at least 1)
    if (G::UNIFORM_ROWS && rows_left >= KT) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        kreg[i] = *(const u32x4*)(c.kp + i * G::ROWS_PER_PASS * k_row_bytes + lane_off_k);
        vreg[i] = *(const u32x4*)(c.vp + i * G::ROWS_PER_PASS * v_row_bytes + lane_off_v);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int cidx = tid + THREADS * i;
        const int row = min(min(cidx / CPR, KT - 1), rows_left - 1), ch = cidx % CPR;
        kreg[i] = *(const u32x4*)(c.kp + (uint32_t)(row * (int)k_row_bytes + ch * 16));
        vreg[i] = *(const u32x4*)(c.vp + (uint32_t)(row * (int)v_row_bytes + ch * 16));
      }
    }
  };
  auto commit = [&](auto& kreg, auto& vreg, int buf, const Cursor& c) {
    const int n = c.u.n, kt = c.kt;
    unsigned char* kimg = smem + buf * BUF;
    unsigned char* vimg = kimg + KBYTES;
    const bool tail = kt * KT + KT > n;  // workgroup-uniform
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int cidx = tid + THREADS * i;
      const int row = cidx / CPR, ch = cidx - row * CPR;
      if (TOTAL % THREADS == 0 || cidx < TOTAL) {
        u32x4 kv = kreg[i], vv = vreg[i];
        if (tail && kt * KT + row >= n) kv = vv = u32x4{0, 0, 0, 0};  // rows past the prompt read as zero
        if constexpr (F32) {
          uint32_t* kp = (uint32_t*)(kimg + row * KSTR + ch * 16);
          kp[0] = kv.x;
          kp[1] = kv.y;
          kp[2] = kv.z;
          kp[3] = kv.w;
        } else {
          *(u32x4*)(kimg + row * KSTR + ch * 16) = kv;
        }
        *(u32x4*)(vimg + row * VSTR + ch * 16) = vv;
      }
    }
  };

  f32x16 o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
  float m = -INFINITY, l = 0.f;

  // Transposed V read: lane 4 qq + p of a 16-lane group addresses row qq,
  // columns 4p..4p+3 of a 4 x 16 block; the group (lane >> 4) selects the
  // 16-column half of the d block (bit 0) and the key quad (bit 1 = hh).
  const int tr_off = (4 * hh + ((lane >> 2) & 3)) * VSTR + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  // one (key block, 16-key step) group g = 2 kb + s2: per 32-row block of O^T two
  // transposed reads (keys 16 s2 + 4 hh + 0..3 and + 8), kept as the low halves
  // of two registers pairs
  auto load_vt = [&](const unsigned char* vimg, int g, u32x4 (&dst)[DB][2]) {
    if constexpr (!F32) {
      const unsigned char* vrow = vimg + tr_off + g * 16 * VSTR;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow + 64 * db));
        const s16x4 hi =
            __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vrow + 64 * db + 8 * VSTR));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        dst[db][0] = u32x4{l2.x, l2.y, 0, 0};
        dst[db][1] = u32x4{h2.x, h2.y, 0, 0};
      }
    }
  };

  // ---- a unit's rows leave -----------------------------------------------------
  auto epilogue = [&](const Unit& it, const int q0) {
    const int lane = opaque_lane();
    const int r = lane & 31, hh = lane >> 5;
    if constexpr (F32) l += other_half(l);  // the 16-bit path's sums already cover both halves
    const float inv = 1.f / l;
    if constexpr (F32) {
      if (q0 + r < it.n) {
        unsigned char* orow = (unsigned char*)a.out + ((it.t0 + q0 + r) * a.num_heads + it.head) * (int64_t)D * ES;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d = 32 * db + 8 * g + 4 * hh;  // registers 4g..4g+3 = rows d..d+3 of O^T
            if (d < D)
              *(f32x4*)(orow + d * 4) = f32x4{o[db][4 * g] * inv, o[db][4 * g + 1] * inv, o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv};
          }
      }
    } else {
      // Registers 4g..4g+3 of a lane are rows 8g + 4hh + 0..3 of the d block: 8 bytes of the output
      // row, the other 8 of the same 16 sit in lane l ^ 32.  Two v_permlane32_swap per pair of
      // groups give lanes 0-31 the 16 bytes of group g and lanes 32-63 those of group g + 1.  The
      // 16-byte pieces go to the wave's LDS strip (memory of its own where the workgroup goes on to
      // another unit; else the K/V images', which are dead once every wave is past the last tile's
      // barrier) and come back with LPR lanes per row: global stores of whole rows instead of 64
      // pieces in 32 rows per instruction.
      constexpr int OSTR = G::OSTR;
      unsigned char* strip = smem + G::STRIP_BASE + wave * 32 * OSTR;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int g = 2 * gp;
          const uint32_t x0 = X::pack2(o[db][4 * g] * inv, o[db][4 * g + 1] * inv);
          const uint32_t x1 = X::pack2(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv);
          const uint32_t y0 = X::pack2(o[db][4 * g + 4] * inv, o[db][4 * g + 5] * inv);
          const uint32_t y1 = X::pack2(o[db][4 * g + 6] * inv, o[db][4 * g + 7] * inv);
          // swap(old, src): old keeps its lanes 0-31 and takes src's lanes 0-31 into its lanes 32-63;
          // src takes old's lanes 32-63 into its lanes 0-31 and keeps its lanes 32-63
          const auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
          const int d = 32 * db + 8 * (g + hh);
          if (d < D) *(u32x4*)(strip + r * OSTR + d * 2) = u32x4{(uint32_t)s0[0], (uint32_t)s1[0], (uint32_t)s0[1], (uint32_t)s1[1]};
        }
      __builtin_amdgcn_wave_barrier();
      constexpr int LPR = CPR <= 8 ? 8 : 16;  // lanes per output row (CPR of them store)
      constexpr int RPI = 64 / LPR;           // rows per store instruction
      const int chunk = lane % LPR;
      unsigned char* obase = (unsigned char*)a.out + ((it.t0 + q0) * a.num_heads + it.head) * (int64_t)D * ES;
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int row = i * RPI + lane / LPR;
        if (chunk < CPR && q0 + row < it.n) {
          const u32x4 v = *(const u32x4*)(strip + row * OSTR + chunk * 16);
          *(u32x4*)(obase + (int64_t)row * a.num_heads * D * ES + chunk * 16) = v;
        }
      }
    }
  };

  // The stream of tiles: ci = the next tile to load, cc = the next tile to commit, (cur, kt) = the tile to
  // compute.  Stream position p lives in LDS buffer p & 1.
  Cursor ci;
  ci.u = load_unit(0);
  ci.j = 0;
  first_tile(ci);
  Cursor cc = ci;
  Unit cur = ci.u;
  int cur_j = 0, kt = 0;
  int q0 = cur.qb * kQBlock + wave * 32;  // this wave's first query row (inside the prompt)
  bool wave_has_rows = q0 < cur.n;
  bool done = false;
  load_q(cur);
  negate_q_if_asked();
  issue(kreg, vreg, ci);
  advance(ci);
  commit(kreg, vreg, 0, cc);
  advance(cc);
  __syncthreads();

  PA_T(t_loop);
  PA_T_ADD(7, t_loop, t_start);
  // One tile: issue the loads of stream position step + AHEAD into (ki, vi), compute position step, commit
  // position step + 1 from (kc, vc); when that was the unit's last tile, send its rows out and take the next unit.
  // The stream position's parity is a compile-time constant of each of the loop's two copies of the step, so
  // that the buffer's place in LDS folds into the instructions' offsets.
  auto tile_step = [&](auto parity, auto& ki, auto& vi, auto& kc, auto& vc) {
    constexpr int buf = decltype(parity)::value;
    PA_T(t0);
    if (ci.u.valid) {
      issue(ki, vi, ci);
      advance(ci);
    }
    PA_T(t1);
    PA_T_ADD(0, t1, t0);

    if (wave_has_rows && kt * KT <= q0 + 31) {  // wave-uniform: tiles above the diagonal are skipped
      const unsigned char* kimg = smem + buf * BUF;
      const unsigned char* vimg = kimg + KBYTES;

      // S^T[key][q]: accumulator register = key row, lane = query row
      f32x16 s[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
      [[maybe_unused]] u32x4 vt[KB * 2][DB][2];  // V^T fragments of one (key block, 16-key step) group
      if constexpr (F32) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const unsigned char* krow = kimg + (32 * kb + r) * KSTR;
#pragma unroll
          for (int ks = 0; ks < D / 2; ++ks) {
            const float kv = *(const float*)(krow + (hh * (D / 2) + ks) * 4);
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv, __builtin_bit_cast(float, qf[ks]), s[kb], 0, 0, 0);
          }
        }
      } else {
        // K fragments are fetched kPrefetch k-steps ahead of the MFMA that
        // consumes them, so the LDS latency overlaps the matrix pipe.
        constexpr int KS = D / 16, NS = KB * KS;
        constexpr int kPrefetch = G::MIN_WAVES > 2 ? PA_PROMPT_PREFETCH_TIGHT : PA_PROMPT_PREFETCH;
        const unsigned char* krow = kimg + r * KSTR + 16 * hh;
        u32x4 kf[NS];
        if constexpr (G::CLUSTER_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int st = 0; st < kPrefetch && st < NS; ++st)
          kf[st] = *(const u32x4*)(krow + (st / KS) * 32 * KSTR + (st % KS) * 32);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          if (st + kPrefetch < NS)
            kf[st + kPrefetch] = *(const u32x4*)(krow + ((st + kPrefetch) / KS) * 32 * KSTR + ((st + kPrefetch) % KS) * 32);
          const int ks = st % KS;
          const u32x4 qv = {qf[4 * ks], qf[4 * ks + 1], qf[4 * ks + 2], qf[4 * ks + 3]};
          s[st / KS] = X::mfma(kf[st], qv, s[st / KS]);
        }
        // the first group of V^T fragments does not depend on the softmax: fetch it now
        load_vt(vimg, 0, vt[0]);
#if PA_PROMPT_SCHED
        __builtin_amdgcn_sched_group_barrier(0x100, kPrefetch, 0);
#pragma unroll
        for (int st = 0; st < NS - kPrefetch; ++st) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, kPrefetch, 0);
#endif
        if constexpr (G::CLUSTER_PRIO) __builtin_amdgcn_s_setprio(0);
      }

      PA_T(t2);
      PA_T_ADD(1, t2, t1);
      // causal mask on the tiles that reach the diagonal (this also hides the
      // zero rows staged past the end of the prompt from every valid query)
      if (kt * KT + KT - 1 > q0) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = kt * KT + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (key > q0 + r) s[kb][i] = -INFINITY;
          }
      }

      // online softmax, log2 domain.  Every processed tile holds at least one
      // unmasked key for every row of the wave, so the tile maximum is finite.
      float mx = s[0][0];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kb][i]);
      mx = fmaxf(mx, other_half(mx)) * a.c;
      // The running maximum m follows the true one lazily: it moves only when
      // some row of the wave outgrows it by more than kDeferLog2, so most tiles
      // skip the rescale of O^T.  Weights then reach 2^kDeferLog2 instead of 1,
      // which costs no relative precision in floating point.
      float alpha = 1.f;
      if (__builtin_amdgcn_ballot_w64(mx - m > kDeferLog2) != 0) {  // wave-uniform (m = -inf at the first tile)
        const float m_new = fmaxf(m, mx);
        alpha = __builtin_amdgcn_exp2f(m - m_new);
        m = m_new;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
      }
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][i], a.c, -m));
          s[kb][i] = p;
          if constexpr (F32) psum += p;
        }
      if constexpr (F32) l = l * alpha + psum;  // per-lane partial sum: lanes l and l+32 are added at the end

      PA_T(t3);
      PA_T_ADD(2, t3, t2);
      // O^T[d][q] += V^T[d][key] P^T[key][q]
      if constexpr (F32) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * hh;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
              const float vv = *(const float*)(vimg + key * VSTR + (32 * db + r) * 4);
              o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, s[kb][i], o[db], 0, 0, 0);
            }
          }
      } else {
        f32x16 lsum;
#pragma unroll
        for (int i = 0; i < 16; ++i) lsum[i] = 0.f;
        const u32x4 ones = {X::kOnes, X::kOnes, X::kOnes, X::kOnes};
        if constexpr (G::CLUSTER_PRIO) __builtin_amdgcn_s_setprio(1);
        // registers 8 s2 .. 8 s2 + 7 of block kb = keys 16 s2 + {4 hh + 0..3, 8 + 4 hh + 0..3}
        auto pack_group = [&](int g) {
          const int kb = g >> 1, s2 = g & 1;
          return u32x4{X::pack2(s[kb][8 * s2 + 0], s[kb][8 * s2 + 1]), X::pack2(s[kb][8 * s2 + 2], s[kb][8 * s2 + 3]),
                       X::pack2(s[kb][8 * s2 + 4], s[kb][8 * s2 + 5]), X::pack2(s[kb][8 * s2 + 6], s[kb][8 * s2 + 7])};
        };
        // Where registers are short (MIN_WAVES > 2) all weights are converted before the first
        // MFMA, which ends the 32 fp32 registers' lives there; else group by group between the MFMAs.
        [[maybe_unused]] u32x4 pf_all[KB * 2];
        if constexpr (G::MIN_WAVES > 2) {
#pragma unroll
          for (int g = 0; g < KB * 2; ++g) pf_all[g] = pack_group(g);
        }
#pragma unroll
        for (int g = 0; g < KB * 2; ++g) {
          if (g + 1 < KB * 2) load_vt(vimg, g + 1, vt[g + 1]);  // next group's fragments behind this group's MFMAs
          u32x4 pf;
          if constexpr (G::MIN_WAVES > 2) pf = pf_all[g]; else pf = pack_group(g);
          // row sums of the rounded weights: a block of ones as the A operand puts
          // sum_key P[key][q] (all 64 lanes' keys) into every register of lsum
          lsum = X::mfma(ones, pf, lsum);
#pragma unroll
          for (int db = 0; db < DB; ++db) {
            const u32x4 vfrag = {vt[g][db][0].x, vt[g][db][0].y, vt[g][db][1].x, vt[g][db][1].y};
            o[db] = X::mfma(vfrag, pf, o[db]);
          }
        }
#if PA_PROMPT_SCHED
        // order pinned for the scheduler: the next group's transposed reads go out
        // two at a time between this group's MFMAs
#pragma unroll
        for (int g = 0; g < KB * 2; ++g) {
          if (g + 1 < KB * 2) {
#pragma unroll
            for (int db = 0; db < DB; ++db) {
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, DB + 1, 0);
          }
        }
#endif
        if constexpr (G::CLUSTER_PRIO) __builtin_amdgcn_s_setprio(0);
        l = l * alpha + lsum[0];
      }
#ifdef PA_PROMPT_TIMING
      asm volatile("s_nop 0" : : "v"(o[0][0]));  // the last MFMA's result has to be there: the phase ends with it
#endif
      PA_T(t4);
      PA_T_ADD(3, t4, t3);
#ifdef PA_PROMPT_TIMING
      phase_acc[6] += 1;
#endif
    }

    PA_T(t5);
    if (cc.u.valid) {
      commit(kc, vc, buf ^ 1, cc);
      advance(cc);
    }
    PA_T(t6);
    PA_T_ADD(4, t6, t5);
    __syncthreads();
    PA_T(t7);
    PA_T_ADD(5, t7, t6);
    if constexpr (!G::PERSIST) {
      done = ++kt == cur.n_tiles;  // the one unit's rows leave after the loop
    } else if (++kt == cur.n_tiles) {
      // The unit's last tile is done.  The next unit's queries are fetched first (the old fragments are
      // dead), so that they arrive while this unit's rows are normalised and stored.  (Fetching them a
      // tile earlier, right after the unit's last scores, needs a second set of registers in the
      // compiler's hands: 366 spilled at head size 128.)
      PA_T(t_epi);
      const Unit old = cur;
      const int old_q0 = q0;
      cur = load_unit(++cur_j);
      if (cur.valid) load_q(cur);
      epilogue(old, old_q0);
      if (cur.valid) negate_q_if_asked();
      // The fetch is waited for HERE, once per unit: left pending across the loop's back edge, every tile's
      // first MFMA would wait for all loads before it, the tile's own K/V loads included (measured: the
      // score phase twice as long).
#pragma unroll
      for (int i = 0; i < NQ; ++i) asm volatile("" : "+v"(qf[i]));
      q0 = cur.qb * kQBlock + wave * 32;
      wave_has_rows = cur.valid && q0 < cur.n;
      kt = 0;
      m = -INFINITY;
      l = 0.f;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
      done = !cur.valid;
#ifdef PA_PROMPT_TIMING
      const unsigned long long t_end = __builtin_amdgcn_s_memtime();
      phase_acc[8] += t_end - t_epi;
#endif
    }
  };
  if constexpr (AHEAD == 2) {
    // position 1 goes into the first set, which the prologue has just emptied; from then on the sets alternate
    // (sending its loads out with position 0's, before the first commit, was measured: +2.8 % on 4 x 4096 but
    // -1 ... -3 % on 512- to 2048-token prompts, whose rounds of workgroups start together and are bound by the
    // bandwidth of that first burst, not by its latency)
    if (ci.u.valid) {
      issue(kreg, vreg, ci);
      advance(ci);
    }
    while (true) {
      tile_step(std::integral_constant<int, 0>{}, kreg2, vreg2, kreg, vreg);
      if (done) break;
      tile_step(std::integral_constant<int, 1>{}, kreg, vreg, kreg2, vreg2);
      if (done) break;
    }
  } else {
    while (true) {
      tile_step(std::integral_constant<int, 0>{}, kreg, vreg, kreg, vreg);
      if (done) break;
      tile_step(std::integral_constant<int, 1>{}, kreg, vreg, kreg, vreg);
      if (done) break;
    }
  }
  if constexpr (!G::PERSIST) {
    PA_T(t_epi);
    epilogue(cur, q0);
#ifdef PA_PROMPT_TIMING
    phase_acc[8] += __builtin_amdgcn_s_memtime() - t_epi;
#endif
  }
#ifdef PA_PROMPT_TIMING
  {
    phase_acc[9] += 1;
    const unsigned slot = blockIdx.x * (unsigned)NW + (unsigned)wave;  // a slot per wave: no atomics, which would serialise
    if (lane == 0 && slot < (unsigned)kTimingWaves)
      for (int i = 0; i < 10; ++i) prompt_phase_cycles[slot][i] = phase_acc[i];
  }
#endif
}

// ---- host side -----------------------------------------------------------------

template <typename X, int D, int NW>
int launch_prompt(PromptArgs& args, int num_work, hipStream_t stream) {
  using G = PromptGeometry<X, D, NW>;
  static bool configured_on[64] = {};  // per device; benign race: the attribute call is idempotent
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(PA_ERR_HIP, "prompt attention: no current device");
  bool& configured = configured_on[dev];
  // tuning builds: extra LDS per workgroup, to keep a second workgroup off the CU
  static const int lds_pad = tuning_int("PA_PROMPT_LDS_PAD", 0);
  const int lds_bytes = G::LDS_BYTES + (NW == 4 ? lds_pad : 0);
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pa_prompt_kernel<X, D, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return fail(PA_ERR_HIP, "prompt attention: hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  args.num_units = num_work * args.num_heads;
  // several units per workgroup: one workgroup per CU (each fills its CU), fewer for a small launch
  const int grid = G::PERSIST ? std::min(args.num_units, kChipCUs) : args.num_units;
  pa_prompt_kernel<X, D, NW><<<dim3((unsigned)grid), dim3(G::THREADS), lds_bytes, stream>>>(args);
  PA_CHECK_LAUNCH("pa_prompt_kernel");
  return PA_OK;
}

template <typename X, int NW>
int dispatch_head(PromptArgs& args, int num_work, int head_size, hipStream_t stream) {
  switch (head_size) {
    case 64: return launch_prompt<X, 64, NW>(args, num_work, stream);
    case 80: return launch_prompt<X, 80, NW>(args, num_work, stream);
    case 96: return launch_prompt<X, 96, NW>(args, num_work, stream);
    case 128: return launch_prompt<X, 128, NW>(args, num_work, stream);
    default:
      return fail(PA_ERR_UNSUPPORTED, "prompt attention: unsupported head size %d (supported: 64, 80, 96, 128)", head_size);
  }
}

template <int NW>
int dispatch_dtype(PromptArgs& args, int num_work, int head_size, int dtype, hipStream_t stream) {
  if (dtype == PA_DTYPE_F16) return dispatch_head<OpsF16, NW>(args, num_work, head_size, stream);
  if (dtype == PA_DTYPE_BF16) return dispatch_head<OpsBF16, NW>(args, num_work, head_size, stream);
  return dispatch_head<OpsF32, NW>(args, num_work, head_size, stream);
}

// Query rows per workgroup.  256-row workgroups (8 waves) stage every K/V tile
// for twice as many rows, which is worth it once the grid still fills the chip
// several times over; smaller batches keep 128 rows for the finer grain.
int choose_waves(const int32_t* cu, int num_seqs, int num_heads) {
#ifdef PA_PROMPT_FORCE_WAVES
  static const int forced = PA_PROMPT_FORCE_WAVES;
#else
  static const int forced = tuning_int("PA_PROMPT_WAVES", 0);  // tuning builds: 4 or 8
#endif
  if (forced == 4 || forced == 8) return forced;
  int64_t big_blocks = 0;
  for (int s = 0; s < num_seqs; ++s) big_blocks += (cu[s + 1] - cu[s] + 255) / 256;
  return big_blocks * num_heads >= 4 * 256 ? 8 : 4;
}

// Heaviest-first alone starts a chip-full of equally long workgroups together whenever the batch
// holds that many (8 prompts x 32 heads: 256 per query-block level), and then every round ends at
// once and the next one begins with all its workgroups fetching their queries and first tiles in
// the same microseconds.  With more than two rounds of work, the first round is instead dealt from
// evenly spaced places of the sorted list, so that its workgroups end at different times and the
// rest (still heaviest first) start one by one.
#ifndef PA_PROMPT_STAGGER
#define PA_PROMPT_STAGGER 1
#endif
void stagger_first_round(std::vector<uint32_t>& work, int num_heads, int waves, int head_size, int dtype) {
  if (PA_PROMPT_PERSISTENT && dtype != PA_DTYPE_F32) return;  // those launches deal the units to resident workgroups themselves
  static const int mode = tuning_int("PA_PROMPT_STAGGER", PA_PROMPT_STAGGER);
  if (mode == 0) return;
  // Only where a workgroup has its CU to itself (8 waves, except at head size 64 with its 128 registers):
  // two workgroups per CU overlap one's start with the other's loop anyway, and there the staggered order
  // only loses the heaviest-first balance (head size 64, 12 heads, 64 x 512: 12 % slower staggered).
  if (waves != 8 || head_size == 64) return;
  const int slots = kChipCUs;
  const size_t first = (size_t)std::max(1, slots / num_heads);  // work items of the first round
  if (work.size() <= 2 * first) return;  // at most two rounds: nothing to stagger against
  // A first round that already holds several different lengths (one or two long prompts: every query block
  // is a level of its own or shared by two) ends staggered by itself, and heaviest-first is then the better
  // order (measured, bf16, 32 heads x 128: 1 x 16384 and 2 x 8192 2 % slower staggered; 4 x 4096 +1 ... 2.5 %,
  // 8 x 2048 +2 ... 5 %, 16 x 1024 +5 ... 7 %, 32 x 512 +4 %).
  size_t levels = 1;
  for (size_t i = 1; i < first; ++i) levels += (work[i] & 0xffffu) != (work[i - 1] & 0xffffu);
  if (levels * 4 > first) return;
  std::vector<uint32_t> head_part, rest;
  std::vector<char> taken(work.size(), 0);
  // mode 1: spaced over the whole list; 2: over its heaviest three quarters; 3: over its heavier half
  // (the lightest items stay at the end, where they even out the finish)
  const size_t span = mode == 2 ? work.size() * 3 / 4 : mode == 3 ? work.size() / 2 : work.size();
  for (size_t i = 0; i < first; ++i) {
    const size_t j = i * span / first;
    taken[j] = 1;
    head_part.push_back(work[j]);
  }
  for (size_t j = 0; j < work.size(); ++j)
    if (!taken[j]) rest.push_back(work[j]);
  work = head_part;
  work.insert(work.end(), rest.begin(), rest.end());
}

}  // namespace

#ifdef PA_PROMPT_TIMING
extern "C" int pa_prompt_timing_read(unsigned long long* out16) {
  static std::vector<unsigned long long> host((size_t)kTimingWaves * 10);
  if (hipMemcpyFromSymbol(host.data(), HIP_SYMBOL(prompt_phase_cycles), host.size() * 8) != hipSuccess) return 1;
  for (int i = 0; i < 16; ++i) out16[i] = 0;
  for (int w = 0; w < kTimingWaves; ++w)
    for (int i = 0; i < 10; ++i) out16[i] += host[(size_t)w * 10 + i];
  void* dev = nullptr;
  if (hipGetSymbolAddress(&dev, HIP_SYMBOL(prompt_phase_cycles)) != hipSuccess) return 1;
  return hipMemset(dev, 0, host.size() * 8) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int pa_multi_query_kv_attention(void* out, const void* query, const void* key, const void* value,
                                           const int32_t* cu_seqlens, int num_seqs, int num_heads, int head_size,
                                           float scale, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                           int dtype, void* stream) {
  clear_error();
  PA_REQUIRE(num_seqs >= 0 && num_heads > 0 && head_size > 0, "prompt attention: bad sizes (num_seqs %d, num_heads %d, head_size %d)",
             num_seqs, num_heads, head_size);
  if (dtype != PA_DTYPE_F32 && dtype != PA_DTYPE_F16 && dtype != PA_DTYPE_BF16)
    return fail(PA_ERR_UNSUPPORTED, "prompt attention: unsupported dtype code %d", dtype);
  if (num_seqs == 0) return PA_OK;
  PA_REQUIRE(cu_seqlens != nullptr, "prompt attention: cu_seqlens is null");
  PA_REQUIRE(isfinite(scale), "prompt attention: scale must be finite");
  const int64_t total = (int64_t)cu_seqlens[num_seqs] - cu_seqlens[0];
  for (int s = 0; s < num_seqs; ++s) {
    PA_REQUIRE(cu_seqlens[s + 1] >= cu_seqlens[s] && cu_seqlens[s] >= 0, "prompt attention: cu_seqlens must be non-decreasing");
    PA_REQUIRE((int64_t)cu_seqlens[s + 1] - cu_seqlens[s] <= (int64_t)65535 * 128, "prompt attention: prompt %d is too long", s);
  }
  if (total == 0) return PA_OK;
  PA_REQUIRE(out && query && key && value, "prompt attention: null tensor pointer");
  const int64_t row = (int64_t)num_heads * head_size;
  PA_REQUIRE(q_stride >= row && k_stride >= row && v_stride >= row, "prompt attention: row strides are smaller than num_heads * head_size");
  const int es = elem_size(dtype);
  if ((k_stride | v_stride) * es >= (1ll << 24))
    return fail(PA_ERR_UNSUPPORTED, "prompt attention: key / value row strides of 16 MiB or more are not supported");
  if (!aligned16(out) || !aligned16(query) || !aligned16(key) || !aligned16(value) || (q_stride * es) % 16 || (k_stride * es) % 16 ||
      (v_stride * es) % 16)
    return fail(PA_ERR_UNSUPPORTED, "prompt attention: tensors and row strides must be 16-byte aligned");

  PromptArgs args;
  args.out = out;
  args.q = query;
  args.k = key;
  args.v = value;
  args.q_stride = q_stride;
  args.k_stride = k_stride;
  args.v_stride = v_stride;
  args.c = fabsf(scale) * 1.4426950408889634f;
  args.q_mode = scale < 0.f ? 1 : 0;
  if (scale == 0.f) {  // uniform weights: zero scores with any positive factor
    args.c = 1.f;
    args.q_mode = 2;
  }
  args.num_heads = num_heads;
  args.num_units = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);

  const int waves = choose_waves(cu_seqlens, num_seqs, num_heads);
  const int qblock = 32 * waves;
  std::vector<uint32_t> work;
  for (int s0 = 0; s0 < num_seqs; s0 += kMaxSeqs) {
    const int ns = std::min(kMaxSeqs, num_seqs - s0);
    for (int i = 0; i <= ns; ++i) args.cu[i] = cu_seqlens[s0 + i];
    work.clear();
    for (int i = 0; i < ns; ++i) {
      const int n = args.cu[i + 1] - args.cu[i];
      for (int qb = 0; qb * qblock < n; ++qb) work.push_back(((uint32_t)i << 16) | (uint32_t)qb);
    }
    // query blocks far down a prompt see the most keys: start them first
    std::stable_sort(work.begin(), work.end(), [](uint32_t x, uint32_t y) { return (x & 0xffffu) > (y & 0xffffu); });
    stagger_first_round(work, num_heads, waves, head_size, dtype);
    for (size_t w0 = 0; w0 < work.size(); w0 += kMaxWork) {
      const int nw = (int)std::min<size_t>(kMaxWork, work.size() - w0);
      std::copy(work.begin() + w0, work.begin() + w0 + nw, args.work);
      const int rc = waves == 8 ? dispatch_dtype<8>(args, nw, head_size, dtype, st)
                                : dispatch_dtype<4>(args, nw, head_size, dtype, st);
      if (rc != PA_OK) return rc;
    }
  }
  return PA_OK;
}
