// KV-cache maintenance kernels for gfx950 (MI355X): reshape_and_cache,
// gather_cached_kv, copy_blocks, swap_blocks.  They replace
// csrc/cache_kernels.cu of vLLM v0.1.0.  All of them are byte movers, so they
// are written over raw 16-byte vectors and element-sized words; the element
// type only enters through its size.  Offsets are 64-bit throughout.
#include <stdlib.h>

#include <vector>

#include "pa_common.h"
#include "pa_exact_math.h"

namespace pa {

// 16-byte cache-side stores carry the non-temporal hint.  Measured on MI355X:
// after default-policy stores into a layer's cache, the next decode kernel over
// that layer runs slower although only one token per sequence was written
// (batch 256 x 32 heads x ctx 1024: 630 us untouched, +22 us after the 16-byte
// K stores, +36 us after the 2-byte V stores, +53 us after both) — the first
// read of lines a previous kernel left in the caches' write path is slow.
// With the hint the stores go to memory and the penalty is gone.  Element-sized
// stores keep the default policy: as non-temporal stores each of them becomes a
// separate fabric write, which makes reshape_and_cache itself 4-6x slower
// (51 vs 13 us for a decode step, 3.2 vs 0.5 ms for a 16k-token prompt).
// 0 is kept as a build-time knob for A/B runs.
#ifndef PA_CACHE_NT_STORE
#define PA_CACHE_NT_STORE 1
#endif

template <typename W>
__device__ __forceinline__ void store_cache(W* p, W v) {
#if PA_CACHE_NT_STORE
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

template <int BYTES> struct Word;
template <> struct Word<1> { typedef uint8_t type; };
template <> struct Word<2> { typedef uint16_t type; };
template <> struct Word<4> { typedef uint32_t type; };
template <> struct Word<8> { typedef u32x2 type; };
template <> struct Word<16> { typedef u32x4 type; };

// ---- reshape_and_cache / gather_cached_kv ------------------------------------
//
// grid = (tokens, slices of the token row): a decode step has only a few
// hundred tokens, so each token's row is cut into slices of one workgroup's
// width to put several workgroups on every CU.  The token's H*D elements are walked in 16-byte
// vectors of X = 16/ES elements: a K vector is exactly one innermost [x] run
// of the key cache (one 16-byte access on both sides); the X elements of a V
// vector land B elements apart in the value cache (element-sized accesses on
// the cache side, one 16-byte access on the token side).
//
// GATHER = false: token rows -> caches (reference reshape_and_cache_kernel,
//                 csrc/cache_kernels.cu:143-182)
// GATHER = true : caches -> token rows (gather_cached_kv_kernel, :230-270)
// VEC = false is the element-wise fallback for token rows that are not
// 16-byte aligned.
template <int ES, bool GATHER, bool VEC>
__global__ void __launch_bounds__(128)
pa_token_cache_kernel(char* __restrict__ key, char* __restrict__ value, char* __restrict__ key_cache,
                      char* __restrict__ value_cache, const int* __restrict__ slot_mapping, const int num_heads,
                      const int head_size, const int block_size, const int64_t key_stride_bytes,
                      const int64_t value_stride_bytes) {
  typedef typename Word<ES>::type elem_t;
  constexpr int X = 16 / ES;
  const int token = blockIdx.x;
  const int slot = slot_mapping[token];
  if (slot < 0) return;
  const int64_t blk = slot / block_size;
  const int off = slot % block_size;
  const int n = num_heads * head_size;
  char* krow = key + (int64_t)token * key_stride_bytes;
  char* vrow = value + (int64_t)token * value_stride_bytes;
  const int64_t cache_block = blk * (int64_t)n * block_size;  // elements

  if (VEC) {
    const int nvec = n / X;
    const int ndx = head_size / X;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < nvec; i += gridDim.y * blockDim.x) {
      const int h = i / ndx;
      const int dx = i - h * ndx;
      // key cache [blk][h][dx][off][x]
      const int64_t kc_el = cache_block + (((int64_t)h * ndx + dx) * block_size + off) * X;
      // value cache [blk][h][d][off], d = dx * X + e
      const int64_t vc_el = cache_block + ((int64_t)h * head_size + (int64_t)dx * X) * block_size + off;
      u32x4* ktok = reinterpret_cast<u32x4*>(krow + (int64_t)i * 16);
      u32x4* vtok = reinterpret_cast<u32x4*>(vrow + (int64_t)i * 16);
      u32x4* kc = reinterpret_cast<u32x4*>(key_cache + kc_el * ES);
      elem_t* vc = reinterpret_cast<elem_t*>(value_cache + vc_el * ES);
      if (!GATHER) {
        const u32x4 kv = *ktok;
        const u32x4 vv = *vtok;
        elem_t tmp[X];
        __builtin_memcpy(tmp, &vv, 16);
        store_cache(kc, kv);
#pragma unroll
        for (int e = 0; e < X; e++) vc[(int64_t)e * block_size] = tmp[e];
      } else {
        const u32x4 kv = *kc;
        elem_t tmp[X];
#pragma unroll
        for (int e = 0; e < X; e++) tmp[e] = vc[(int64_t)e * block_size];
        u32x4 vv;
        __builtin_memcpy(&vv, tmp, 16);
        *ktok = kv;
        *vtok = vv;
      }
    }
  } else {
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) {
      const int h = i / head_size;
      const int d = i - h * head_size;
      const int64_t kc_el = cache_block + ((((int64_t)h * (head_size / X) + d / X) * block_size + off) * X) + d % X;
      const int64_t vc_el = cache_block + ((int64_t)h * head_size + d) * block_size + off;
      elem_t* ktok = reinterpret_cast<elem_t*>(krow) + i;
      elem_t* vtok = reinterpret_cast<elem_t*>(vrow) + i;
      elem_t* kc = reinterpret_cast<elem_t*>(key_cache) + kc_el;
      elem_t* vc = reinterpret_cast<elem_t*>(value_cache) + vc_el;
      if (!GATHER) {
        *kc = *ktok;
        *vc = *vtok;
      } else {
        *ktok = *kc;
        *vtok = *vc;
      }
    }
  }
}

// Vectors [lo, hi) of a block's value tile, read, given the new token's elements and written back
// whole, U vectors in flight per lane.  Every load is unconditional (indices past the end are
// clamped), the new elements' included: fetched only where needed, each of them waited for its own
// trip to memory between one vector's store and the next (three trips more per workgroup).
// The first pass is split in two — load_first() / finish() — so that a kernel can put the loads in
// flight as soon as it knows the block, and do whatever else it has to wait for (the scan of
// slot_mapping for a second token in the block, the key rows, a rotation) under them.
// Build-time knobs for A/B runs: cache policy of the tile's loads and stores (1 = non-temporal).
#ifndef PA_CACHE_TILE_LOAD_NT
#define PA_CACHE_TILE_LOAD_NT 1
#endif
#ifndef PA_CACHE_TILE_STORE_NT
#define PA_CACHE_TILE_STORE_NT 1
#endif
template <int ES, int U>
struct ValueTileUpdate {
  typedef typename Word<ES>::type elem_t;
  u32x4* tile;
  const elem_t* new_row;
  int lo, hi, vpr, piece, lane_el;
  u32x4 v[U];
  elem_t nv[U];

  __device__ __forceinline__ void load(int i0) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int ic = min(i0 + u * (int)blockDim.x, hi - 1);
#if PA_CACHE_TILE_LOAD_NT
      v[u] = __builtin_nontemporal_load(tile + ic);
#else
      v[u] = tile[ic];
#endif
      nv[u] = new_row[ic / vpr];
    }
  }
  __device__ __forceinline__ void patch_and_store(int i0) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = i0 + u * blockDim.x;
      if (i >= hi) continue;
      const int r = i / vpr;
      if (i - r * vpr == piece) {
        if (ES == 4) {
          v[u][lane_el] = (uint32_t)nv[u];
        } else {
          const int w = lane_el >> 1, sh = (lane_el & 1) * 16;
          v[u][w] = (v[u][w] & ~(0xffffu << sh)) | ((uint32_t)nv[u] << sh);
        }
      }
#if PA_CACHE_TILE_STORE_NT
      __builtin_nontemporal_store(v[u], tile + i);
#else
      tile[i] = v[u];
#endif
    }
  }
  // lo < hi is the caller's business (an empty slice loads nothing it may not: index hi - 1 >= lo)
  __device__ __forceinline__ void load_first() {
    if (lo < hi) load(lo + (int)threadIdx.x);
  }
  __device__ __forceinline__ void finish() {
    int i0 = lo + (int)threadIdx.x;
    if (lo < hi) patch_and_store(i0);
    for (i0 += U * (int)blockDim.x; i0 - (int)threadIdx.x < hi; i0 += U * (int)blockDim.x) {
      load(i0);
      patch_and_store(i0);
    }
  }
};

// ---- reshape_and_cache for decode-sized batches: whole value rows ------------------
//
// A decode step adds one token to the last block of every sequence.  In the
// value cache [blk][h][d][off] that is one element in each of the block's H * D
// rows of B elements: written as elements (the kernel above), a million 2-byte
// stores per layer, each dirtying part of a 32-byte sector.  The caches write
// those back as masked partial writes while the decode kernel that follows is
// streaming, which costs it 40 us at batch 256 (676 vs 637 us).  Here the
// workgroups of a token read the block's whole value tile (H * D rows, one
// contiguous run of bytes), put the new elements in, and write the tile back
// as whole 16-byte vectors with the streaming hint: twice 128 KiB per token
// instead of 8 KiB, but as full lines that go straight to memory.
//
// Read-modify-write of a block is only safe if no other token of the batch
// goes to the same block (tokens of one prompt do): every workgroup checks the
// whole slot_mapping first and falls back to element stores otherwise, which is
// also what a batch with a repeated slot gets.
template <int ES, int U>
__global__ void __launch_bounds__(256)
pa_token_append_kernel(const char* __restrict__ key, const char* __restrict__ value, char* __restrict__ key_cache,
                       char* __restrict__ value_cache, const int* __restrict__ slot_mapping, const int num_tokens,
                       const int num_heads, const int head_size, const int block_size,
                       const int64_t key_stride_bytes, const int64_t value_stride_bytes) {
  typedef typename Word<ES>::type elem_t;
  constexpr int X = 16 / ES;
  const int token = blockIdx.x;
  const int slot = slot_mapping[token];
  if (slot < 0) return;
  const int64_t blk = slot / block_size;
  const int off = slot % block_size;
  const int n = num_heads * head_size;
  const char* krow = key + (int64_t)token * key_stride_bytes;
  const char* vrow = value + (int64_t)token * value_stride_bytes;
  const int64_t cache_block = blk * (int64_t)n * block_size;  // elements

  // values: the tile is n rows of block_size elements = n * vpr vectors; the
  // new element of row r sits in vector `piece` of the row, at element `lane_el`.
  // Its loads go out first: reading the tile is harmless even if the scan below finds a second
  // token in the block, and everything else this workgroup waits for arrives under them.
  ValueTileUpdate<ES, U> vt;
  vt.vpr = block_size / X;  // vectors per row (>= 1: checked by the host)
  vt.piece = off / X;
  vt.lane_el = off % X;
  vt.tile = reinterpret_cast<u32x4*>(value_cache + cache_block * ES);
  vt.new_row = reinterpret_cast<const elem_t*>(vrow);
  const int total = n * vt.vpr;
  const int per = (total + gridDim.y - 1) / gridDim.y;
  vt.lo = per * blockIdx.y;
  vt.hi = min(vt.lo + per, total);
  vt.load_first();

  // keys: one 16-byte vector of the token row = one innermost [x] run of the key cache
  const int nvec = n / X;
  const int ndx = head_size / X;
  const int kstep = gridDim.y * blockDim.x;
  const int k0 = blockIdx.y * blockDim.x + threadIdx.x;
  u32x4 kfirst = {0u, 0u, 0u, 0u};
  if (k0 < nvec) kfirst = *reinterpret_cast<const u32x4*>(krow + (int64_t)k0 * 16);

  int clash = 0;
  for (int t = threadIdx.x; t < num_tokens; t += blockDim.x) {
    const int s = slot_mapping[t];
    clash |= (t != token && s >= 0 && s / block_size == blk) ? 1 : 0;
  }
  clash = __syncthreads_or(clash);

  for (int i = k0; i < nvec; i += kstep) {
    const int h = i / ndx;
    const int dx = i - h * ndx;
    const int64_t kc_el = cache_block + (((int64_t)h * ndx + dx) * block_size + off) * X;
    const u32x4 kv = i == k0 ? kfirst : *reinterpret_cast<const u32x4*>(krow + (int64_t)i * 16);
    store_cache(reinterpret_cast<u32x4*>(key_cache + kc_el * ES), kv);
  }
  if (clash) {
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x)
      reinterpret_cast<elem_t*>(value_cache)[cache_block + (int64_t)i * block_size + off] =
          reinterpret_cast<const elem_t*>(vrow)[i];
    return;
  }
  vt.finish();
}

// ---- rotary embedding + reshape_and_cache in one pass (decode-sized batches) ---------
//
// In a decoder layer with rotary position embedding the new token's key is rotated in place
// and then copied into the key cache: two kernels, the second re-reading what the first just
// wrote (vllm/model_executor/layers/attention.py:200-206, then :134-140).  Here the workgroups of
// a token rotate its query row in place and its key row in place AND into the cache, from one
// read; the values go the way of pa_token_append_kernel.  Same arithmetic as
// pa_rotary_neox_kernel (every product and sum rounded to T), so query, key and both caches end
// up bit-identical to the two separate calls.  A token without a slot (negative) is rotated and
// not cached, as the two calls would do.
//
// Per head the D / X vectors of a row are: embed / X "x" vectors, their embed / X partners
// ("y", rot_dim / 2 elements further on), and (D - rot_dim) / X vectors the rotation leaves
// alone.  One work item = an (x, y) pair, or one untouched vector.
template <typename T, int U>
__global__ void __launch_bounds__(256)
pa_rotary_append_kernel(const int64_t* __restrict__ positions, T* __restrict__ query, T* __restrict__ key,
                        const T* __restrict__ value, const T* __restrict__ cos_sin_cache, char* __restrict__ key_cache,
                        char* __restrict__ value_cache, const int* __restrict__ slot_mapping, const int num_tokens,
                        const int num_heads, const int head_size, const int rot_dim, const int block_size,
                        const int64_t query_stride, const int64_t key_stride, const int64_t value_stride) {
  constexpr int ES = (int)sizeof(T);
  constexpr int X = 16 / ES;
  typedef Pack<T, X> P;
  typedef typename Word<ES>::type elem_t;
  const int token = blockIdx.x;
  const int slot = slot_mapping[token];
  const int64_t blk = slot >= 0 ? slot / block_size : 0;
  const int off = slot >= 0 ? slot % block_size : 0;
  const int n = num_heads * head_size;
  T* qrow = query + (int64_t)token * query_stride;
  T* krow = key + (int64_t)token * key_stride;
  const T* vrow = value + (int64_t)token * value_stride;
  const T* cs = cos_sin_cache + positions[token] * rot_dim;
  const int64_t cache_block = blk * (int64_t)n * block_size;  // elements

  // values: the block's tile is read first (see pa_token_append_kernel); the scan and the rotation
  // run under its loads
  ValueTileUpdate<ES, U> vt;
  vt.vpr = block_size / X;
  vt.piece = off / X;
  vt.lane_el = off % X;
  vt.tile = reinterpret_cast<u32x4*>(value_cache + cache_block * ES);
  vt.new_row = reinterpret_cast<const elem_t*>(vrow);
  {
    const int total = n * vt.vpr;
    const int per = (total + gridDim.y - 1) / gridDim.y;
    vt.lo = per * blockIdx.y;
    vt.hi = min(vt.lo + per, total);
  }
  if (slot >= 0) vt.load_first();

  int clash = 0;
  if (slot >= 0) {
    for (int t = threadIdx.x; t < num_tokens; t += blockDim.x) {
      const int s = slot_mapping[t];
      clash |= (t != token && s >= 0 && s / block_size == blk) ? 1 : 0;
    }
  }
  clash = __syncthreads_or(clash);
  const int embed = rot_dim / 2;
  const int ndx = head_size / X;   // vectors per head
  const int epx = embed / X;       // (x, y) pairs per head
  const int per_head = ndx - epx;  // work items per head
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < num_heads * per_head; i += gridDim.y * blockDim.x) {
    const int h = i / per_head;
    const int j = i - h * per_head;
    T* kb = krow + (int64_t)h * head_size;
    // key cache [blk][h][dx][off][x]: vector dx of head h
    auto kc_vec = [&](int dx) {
      return reinterpret_cast<u32x4*>(key_cache + (cache_block + (((int64_t)h * ndx + dx) * block_size + off) * X) * ES);
    };
    if (j < epx) {
      const int xo = j * X;
      const P c = *reinterpret_cast<const P*>(cs + xo);
      const P sn = *reinterpret_cast<const P*>(cs + embed + xo);
      T* qb = qrow + (int64_t)h * head_size;
      const P qx = *reinterpret_cast<const P*>(qb + xo), qy = *reinterpret_cast<const P*>(qb + embed + xo);
      const P kx = *reinterpret_cast<const P*>(kb + xo), ky = *reinterpret_cast<const P*>(kb + embed + xo);
      P oqx, oqy, okx, oky;
#pragma unroll
      for (int e = 0; e < X; e++) {
        oqx.v[e] = sub_t(mul_t(qx.v[e], c.v[e]), mul_t(qy.v[e], sn.v[e]));
        oqy.v[e] = add_t(mul_t(qy.v[e], c.v[e]), mul_t(qx.v[e], sn.v[e]));
        okx.v[e] = sub_t(mul_t(kx.v[e], c.v[e]), mul_t(ky.v[e], sn.v[e]));
        oky.v[e] = add_t(mul_t(ky.v[e], c.v[e]), mul_t(kx.v[e], sn.v[e]));
      }
      *reinterpret_cast<P*>(qb + xo) = oqx;
      *reinterpret_cast<P*>(qb + embed + xo) = oqy;
      *reinterpret_cast<P*>(kb + xo) = okx;
      *reinterpret_cast<P*>(kb + embed + xo) = oky;
      if (slot >= 0) {
        store_cache(kc_vec(j), __builtin_bit_cast(u32x4, okx));
        store_cache(kc_vec(j + epx), __builtin_bit_cast(u32x4, oky));
      }
    } else if (slot >= 0) {
      const int dx = j + epx;  // beyond the rotated part
      store_cache(kc_vec(dx), *reinterpret_cast<const u32x4*>(kb + dx * X));
    }
  }
  if (slot < 0) return;
  if (clash) {
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x)
      reinterpret_cast<elem_t*>(value_cache)[cache_block + (int64_t)i * block_size + off] =
          reinterpret_cast<const elem_t*>(vrow)[i];
    return;
  }
  // values: put the new elements into the tile and write it back (see pa_token_append_kernel)
  vt.finish();
}

// ---- reshape_and_cache / gather_cached_kv for prompt-sized batches --------------
//
// A prompt's tokens take consecutive slots, so 16 consecutive tokens fill whole
// rows of the caches: 16 x 16 B of every key [h][dx] row and B elements of every
// value [h][d] row.  The kernel above writes those rows one token at a time (a
// 2-byte value store per element, each its own 32-byte sector write); this one
// stages 16 tokens x `hs` heads in LDS and writes rows:
//   keys    lane <-> (token, (h, dx)), token fastest: the 16 lanes of one (h, dx)
//           cover 256 contiguous bytes when the slots are consecutive;
//   values  lane <-> (h, d) row; the workgroup walks the chunk's runs of
//           consecutive slots inside one block (wave-uniform: every lane sees the
//           same slots) and writes each run's elements of its row with the widest
//           aligned stores (16 / 8 / 4 / ES bytes).
// Nothing is assumed about slot_mapping: isolated tokens are runs of one and
// negative slots are skipped, exactly as in the per-token kernel.  GATHER runs
// the same walk the other way: cache rows -> LDS -> token rows.
constexpr int kRunTokens = 16;
constexpr int kAppendMinTokens = 16;  // decode batches from this size on rewrite whole value tiles

template <int ES, bool GATHER>
__global__ void __launch_bounds__(256)
pa_token_runs_kernel(char* __restrict__ key, char* __restrict__ value, char* __restrict__ key_cache,
                     char* __restrict__ value_cache, const int* __restrict__ slot_mapping, const int num_tokens,
                     const int num_heads, const int head_size, const int block_size, const int64_t key_stride_bytes,
                     const int64_t value_stride_bytes, const int hs) {
  typedef typename Word<ES>::type elem_t;
  constexpr int X = 16 / ES;
  constexpr int TC = kRunTokens;
  extern __shared__ __align__(16) char tile[];
  __shared__ int slots[TC];
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * TC;
  const int h0 = blockIdx.y * hs;
  const int nh = min(hs, num_heads - h0);
  const int rowb = nh * head_size * ES;  // bytes of one token in a tile
  char* ktile = tile;
  char* vtile = tile + TC * hs * head_size * ES;
  const int vec_per_tok = rowb / 16;

  if (tid < TC) slots[tid] = t0 + tid < num_tokens ? slot_mapping[t0 + tid] : -1;
  // token rows <-> LDS tiles, 16-byte vectors, coalesced on the token side
  auto token_rows = [&](bool to_tile) {
    for (int i = tid; i < TC * vec_per_tok; i += 256) {
      const int tok = i / vec_per_tok, c = i - tok * vec_per_tok;
      if (t0 + tok >= num_tokens) continue;
      const int64_t at = (int64_t)h0 * head_size * ES + (int64_t)c * 16;
      u32x4* kg = reinterpret_cast<u32x4*>(key + (int64_t)(t0 + tok) * key_stride_bytes + at);
      u32x4* vg = reinterpret_cast<u32x4*>(value + (int64_t)(t0 + tok) * value_stride_bytes + at);
      u32x4* kl = reinterpret_cast<u32x4*>(ktile + tok * rowb + c * 16);
      u32x4* vl = reinterpret_cast<u32x4*>(vtile + tok * rowb + c * 16);
      if (to_tile) {
        *kl = *kg;
        *vl = *vg;
      } else if (slots[tok] >= 0) {  // tokens without a slot keep their rows
        *kg = *kl;
        *vg = *vl;
      }
    }
  };
  if (!GATHER) token_rows(true);
  __syncthreads();

  // key cache [blk][h][dx][off][x]
  const int ndx = head_size / X;
  for (int i = tid; i < TC * nh * ndx; i += 256) {
    const int tok = i % TC, hd = i / TC;
    const int slot = slots[tok];
    if (slot < 0) continue;
    const int h = hd / ndx, dx = hd - h * ndx;
    const int64_t blk = slot / block_size;
    const int off = slot % block_size;
    u32x4* kl = reinterpret_cast<u32x4*>(ktile + tok * rowb + (h * head_size + dx * X) * ES);
    u32x4* kc = reinterpret_cast<u32x4*>(key_cache) + ((blk * num_heads + h0 + h) * ndx + dx) * block_size + off;
    if (GATHER) *kl = *kc;
    else store_cache(kc, *kl);
  }

  // value cache [blk][h][d][off]
  for (int row = tid; row < nh * head_size; row += 256) {
    const int h = row / head_size, d = row - h * head_size;
    int i = 0;
    while (i < TC) {
      const int slot = slots[i];
      if (slot < 0) {
        ++i;
        continue;
      }
      const int off = slot % block_size;
      int n = 1;
      while (i + n < TC && off + n < block_size && slots[i + n] == slot + n) ++n;
      const int64_t blk = slot / block_size;
      char* crow = value_cache + ((((blk * num_heads + h0 + h) * head_size + d) * block_size) + off) * (int64_t)ES;
      char* lrow = vtile + i * rowb + row * ES;  // token i + j of this row sits at lrow + j * rowb
      int j = 0;
      while (j < n) {
        const int at = (off + j) * ES, rem = (n - j) * ES;
        // widest aligned piece of the cache row: 16, 8, 4 (16-bit types) or ES bytes
        const int bytes = (!(at & 15) && rem >= 16) ? 16 : (!(at & 7) && rem >= 8) ? 8 : (ES == 2 && !(at & 3) && rem >= 4) ? 4 : ES;
        const int cnt = bytes / ES;
        elem_t tmp[X];
        if (GATHER) {
          if (bytes == 16) __builtin_memcpy(tmp, crow + j * ES, 16);
          else if (bytes == 8) __builtin_memcpy(tmp, crow + j * ES, 8);
          else if (bytes == 4) __builtin_memcpy(tmp, crow + j * ES, 4);
          else tmp[0] = *reinterpret_cast<const elem_t*>(crow + j * ES);
#pragma unroll
          for (int e = 0; e < X; e++)
            if (e < cnt) *reinterpret_cast<elem_t*>(lrow + (j + e) * rowb) = tmp[e];
        } else {
#pragma unroll
          for (int e = 0; e < X; e++)
            if (e < cnt) tmp[e] = *reinterpret_cast<const elem_t*>(lrow + (j + e) * rowb);
          if (bytes == 16) {
            u32x4 v;
            __builtin_memcpy(&v, tmp, 16);
            *reinterpret_cast<u32x4*>(crow + j * ES) = v;
          } else if (bytes == 8) {
            u32x2 v;
            __builtin_memcpy(&v, tmp, 8);
            *reinterpret_cast<u32x2*>(crow + j * ES) = v;
          } else if (bytes == 4) {
            uint32_t v;
            __builtin_memcpy(&v, tmp, 4);
            *reinterpret_cast<uint32_t*>(crow + j * ES) = v;
          } else {
            *reinterpret_cast<elem_t*>(crow + j * ES) = tmp[0];
          }
        }
        j += cnt;
      }
      i += n;
    }
  }
  if (GATHER) {
    __syncthreads();
    token_rows(false);
  }
}

// Token count from which reshape_and_cache / gather_cached_kv use the run-aware kernel: decode
// steps (one token per sequence, at most a few hundred) keep the per-token
// kernel, prompt batches of this size or more are mostly long runs.
#ifndef PA_CACHE_RUNS_MIN_TOKENS
#define PA_CACHE_RUNS_MIN_TOKENS 512
#endif

// Launch shape of the whole-tile value update (pa_token_append_kernel, pa_rotary_append_kernel):
// `slices` workgroups per token, each reading and rewriting its share of the block's value tile
// with `in_flight` 16-byte vectors per lane in the air.
struct AppendShape {
  int slices;
  int in_flight;
};
// Measured on MI355X (32 heads x 128, block 16, fp16; reshape_and_cache per launch, graph replay over
// rotating layers; vectors in flight x workgroups): 256 tokens 4 x 2048: 24.4 us, 4 x 1024 (two passes):
// 25.1, 8 x 1024: 22.8, 16 x 512: 23.7; 64 tokens 4: 6.2, 8: 6.4, 16: 7.7; 16 tokens 4: 4.9, 8: 5.6, 16: 7.2.
// Big batches want more bytes in flight per lane, small ones more workgroups.
constexpr int kAppendWideTokens = 128;  // batches from this size on keep 8 vectors in flight per lane
#ifndef PA_CACHE_APPEND_WORKGROUPS
#define PA_CACHE_APPEND_WORKGROUPS 2048
#endif
static AppendShape append_shape(int num_tokens, int tile_vecs) {
  static const int forced = tuning_int("PA_CACHE_APPEND_IN_FLIGHT", 0);  // tuning builds: 4, 8 or 16
  static const int target = tuning_int("PA_CACHE_APPEND_WORKGROUPS", PA_CACHE_APPEND_WORKGROUPS);
  AppendShape sh;
  sh.in_flight = forced == 16 ? 16 : forced == 8 ? 8 : forced == 4 ? 4 : (num_tokens >= kAppendWideTokens ? 8 : 4);
  // about `target` workgroups in all; a slice is at least one pass of in_flight vectors per lane
  int sl = (target + num_tokens - 1) / num_tokens;
  const int most = (tile_vecs + sh.in_flight * 256 - 1) / (sh.in_flight * 256);
  if (sl > most) sl = most;
  if (sl < 1) sl = 1;
  sh.slices = sl;
  return sh;
}

template <bool GATHER>
static int token_cache_op(const char* what, void* key, void* value, void* key_cache, void* value_cache,
                          const int32_t* slot_mapping, int num_tokens, int num_heads, int head_size,
                          int block_size, int64_t key_stride, int64_t value_stride, int dtype, void* stream) {
  clear_error();
  if (dtype != PA_DTYPE_F32 && dtype != PA_DTYPE_F16 && dtype != PA_DTYPE_BF16)
    return fail(PA_ERR_UNSUPPORTED, "%s: unsupported data type code: %d", what, dtype);
  PA_REQUIRE(num_tokens >= 0 && num_heads >= 0 && head_size >= 0 && block_size > 0,
             "%s: negative size or non-positive block size", what);
  if (num_tokens == 0 || num_heads == 0 || head_size == 0) return PA_OK;
  PA_REQUIRE(key && value && key_cache && value_cache && slot_mapping, "%s: null pointer argument", what);
  const int es = elem_size(dtype);
  const int x = 16 / es;
  PA_REQUIRE(head_size % x == 0, "%s: head_size %d is not a multiple of x = %d", what, head_size, x);
  PA_REQUIRE(aligned16(key_cache), "%s: key_cache must be 16-byte aligned", what);
  PA_REQUIRE((int64_t)num_heads * head_size < (1ll << 31), "%s: num_heads * head_size too large", what);
  const bool vec = aligned16(key) && aligned16(value) && (key_stride * es) % 16 == 0 && (value_stride * es) % 16 == 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int work = vec ? num_heads * head_size / x : num_heads * head_size;
  const int threads = work <= 64 ? 64 : 128;
  int slices = (work + threads - 1) / threads;
  // enough workgroups for ~8 per CU at small token counts, one slice per
  // workgroup at most; large prompts already have one row per workgroup
  const int want = (2048 + num_tokens - 1) / num_tokens;
  if (slices > want) slices = want;
  if (slices > 65535) slices = 65535;
  char* k = static_cast<char*>(key);
  char* v = static_cast<char*>(value);
  char* kc = static_cast<char*>(key_cache);
  char* vc = static_cast<char*>(value_cache);
  if (vec && num_tokens >= PA_CACHE_RUNS_MIN_TOKENS && aligned16(value_cache) && (block_size * es) % 16 == 0) {
    int hs = 256 / head_size;
    if (hs < 1) hs = 1;
    if (hs > num_heads) hs = num_heads;
    const size_t lds = (size_t)2 * kRunTokens * hs * head_size * es;
    if (lds <= 48 * 1024) {
      const dim3 grid((num_tokens + kRunTokens - 1) / kRunTokens, (num_heads + hs - 1) / hs);
      if (es == 2)
        hipLaunchKernelGGL((pa_token_runs_kernel<2, GATHER>), grid, dim3(256), lds, st, k, v, kc, vc, slot_mapping,
                           num_tokens, num_heads, head_size, block_size, key_stride * es, value_stride * es, hs);
      else
        hipLaunchKernelGGL((pa_token_runs_kernel<4, GATHER>), grid, dim3(256), lds, st, k, v, kc, vc, slot_mapping,
                           num_tokens, num_heads, head_size, block_size, key_stride * es, value_stride * es, hs);
      PA_CHECK_LAUNCH(what);
      return PA_OK;
    }
  }
  static const int append_rows = tuning_int("PA_CACHE_APPEND_ROWS", 1);  // tuning builds: 0 = element stores for the values
  if (!GATHER && append_rows != 0 && vec && num_tokens >= kAppendMinTokens && aligned16(value_cache) &&
      (block_size * es) % 16 == 0 && (head_size * es) % 16 == 0) {
    const int tile_vecs = num_heads * head_size * (block_size / x);
    const AppendShape sh = append_shape(num_tokens, tile_vecs);
#define PA_LAUNCH_APPEND(ES, U)                                                                                     \
  hipLaunchKernelGGL((pa_token_append_kernel<ES, U>), dim3(num_tokens, sh.slices), dim3(256), 0, st, k, v, kc, vc, \
                     slot_mapping, num_tokens, num_heads, head_size, block_size, key_stride * ES, value_stride * ES)
    if (es == 2) {
      if (sh.in_flight == 16) PA_LAUNCH_APPEND(2, 16); else if (sh.in_flight == 8) PA_LAUNCH_APPEND(2, 8); else PA_LAUNCH_APPEND(2, 4);
    } else {
      if (sh.in_flight == 16) PA_LAUNCH_APPEND(4, 16); else if (sh.in_flight == 8) PA_LAUNCH_APPEND(4, 8); else PA_LAUNCH_APPEND(4, 4);
    }
#undef PA_LAUNCH_APPEND
    PA_CHECK_LAUNCH(what);
    return PA_OK;
  }
#define PA_LAUNCH_TOKEN(ES, VEC)                                                                              \
  hipLaunchKernelGGL((pa_token_cache_kernel<ES, GATHER, VEC>), dim3(num_tokens, slices), dim3(threads), 0, st, k, v, \
                     kc, vc, slot_mapping, num_heads, head_size, block_size, key_stride * ES, value_stride * ES)
  if (es == 2) {
    if (vec) PA_LAUNCH_TOKEN(2, true); else PA_LAUNCH_TOKEN(2, false);
  } else {
    if (vec) PA_LAUNCH_TOKEN(4, true); else PA_LAUNCH_TOKEN(4, false);
  }
#undef PA_LAUNCH_TOKEN
  PA_CHECK_LAUNCH(what);
  return PA_OK;
}

template <typename T>
static void launch_rotary_append(const int64_t* positions, void* query, void* key, const void* value,
                                 const void* cos_sin_cache, void* key_cache, void* value_cache,
                                 const int32_t* slot_mapping, int num_tokens, int num_heads, int head_size, int rot_dim,
                                 int block_size, int64_t query_stride, int64_t key_stride, int64_t value_stride,
                                 hipStream_t st) {
  constexpr int X = 16 / (int)sizeof(T);
  const AppendShape sh = append_shape(num_tokens, num_heads * head_size * (block_size / X));
#define PA_LAUNCH_ROTARY_APPEND(U)                                                                                   \
  hipLaunchKernelGGL((pa_rotary_append_kernel<T, U>), dim3(num_tokens, sh.slices), dim3(256), 0, st, positions,     \
                     static_cast<T*>(query), static_cast<T*>(key), static_cast<const T*>(value),                      \
                     static_cast<const T*>(cos_sin_cache), static_cast<char*>(key_cache), static_cast<char*>(value_cache), \
                     slot_mapping, num_tokens, num_heads, head_size, rot_dim, block_size, query_stride, key_stride,  \
                     value_stride)
  if (sh.in_flight == 16) PA_LAUNCH_ROTARY_APPEND(16); else if (sh.in_flight == 8) PA_LAUNCH_ROTARY_APPEND(8); else PA_LAUNCH_ROTARY_APPEND(4);
#undef PA_LAUNCH_ROTARY_APPEND
}

// ---- copy_blocks ---------------------------------------------------------------
//
// The layer pointer table and the (src, dst) pairs ride in the kernel
// arguments (a few KB), so the host neither allocates device memory nor waits
// for a copy — the reference does three blocking host-to-device copies per
// call (csrc/cache_kernels.cu:116-121).  Larger tables are cut into several
// launches.  grid = (pairs, layers, 2 * splits): z < splits copies a slice of
// the key block, z >= splits the same slice of the value block.
constexpr int kCopyMaxLayers = 96;
constexpr int kCopyMaxPairs = 256;

struct CopyTable {
  char* key_ptr[kCopyMaxLayers];
  char* value_ptr[kCopyMaxLayers];
  int32_t src[kCopyMaxPairs];
  int32_t dst[kCopyMaxPairs];
};

template <int W>
__global__ void __launch_bounds__(256)
pa_copy_blocks_kernel(const CopyTable tab, const int64_t block_bytes, const int splits) {
  typedef typename Word<W>::type word_t;
  const int pair = blockIdx.x, layer = blockIdx.y;
  const int z = blockIdx.z;
  const bool is_value = z >= splits;
  const int part = is_value ? z - splits : z;
  char* base = is_value ? tab.value_ptr[layer] : tab.key_ptr[layer];
  const word_t* __restrict__ s = reinterpret_cast<const word_t*>(base + (int64_t)tab.src[pair] * block_bytes);
  word_t* __restrict__ d = reinterpret_cast<word_t*>(base + (int64_t)tab.dst[pair] * block_bytes);
  const int64_t nwords = block_bytes / W;
  const int64_t per = (nwords + splits - 1) / splits;
  const int64_t lo = per * part;
  const int64_t hi = lo + per < nwords ? lo + per : nwords;
  int64_t i = lo + threadIdx.x;
  // four independent 16-byte loads in flight per lane before the first store
  for (; i + 3 * 256 < hi; i += 4 * 256) {
    const word_t a = s[i], b = s[i + 256], c = s[i + 512], e = s[i + 768];
    store_cache(&d[i], a);
    store_cache(&d[i + 256], b);
    store_cache(&d[i + 512], c);
    store_cache(&d[i + 768], e);
  }
  for (; i < hi; i += 256) store_cache(&d[i], s[i]);
}

// swap_blocks through a kernel: pinned host memory is mapped into the device's
// address space, so one launch can move up to kCopyMaxPairs scattered blocks over
// PCIe / xGMI instead of one hipMemcpyAsync per run of blocks (the host-side cost
// of thousands of small copies is what bounds a swap of scattered blocks).
// grid = (pairs, splits).
struct SwapTable {
  const char* src;
  char* dst;
  int64_t src_block[kCopyMaxPairs];
  int64_t dst_block[kCopyMaxPairs];
};

__global__ void __launch_bounds__(256) pa_swap_blocks_kernel(const SwapTable tab, const int64_t block_bytes, const int splits) {
  const int pair = blockIdx.x, part = blockIdx.y;
  const u32x4* __restrict__ s = reinterpret_cast<const u32x4*>(tab.src + tab.src_block[pair] * block_bytes);
  u32x4* __restrict__ d = reinterpret_cast<u32x4*>(tab.dst + tab.dst_block[pair] * block_bytes);
  const int64_t nwords = block_bytes / 16;
  const int64_t per = (nwords + splits - 1) / splits;
  const int64_t lo = per * part;
  const int64_t hi = lo + per < nwords ? lo + per : nwords;
  int64_t i = lo + threadIdx.x;
  for (; i + 3 * 256 < hi; i += 4 * 256) {
    const u32x4 a = s[i], b = s[i + 256], c = s[i + 512], e = s[i + 768];
    d[i] = a;
    d[i + 256] = b;
    d[i + 512] = c;
    d[i + 768] = e;
  }
  for (; i < hi; i += 256) d[i] = s[i];
}

// Device-visible alias of a pointer that a kernel is about to dereference:
// device memory as it is, host memory only if it is pinned and mapped.
static char* device_alias(const void* p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // unregistered (pageable) host memory
    return nullptr;
  }
  if (attr.type == hipMemoryTypeDevice) return static_cast<char*>(const_cast<void*>(p));
  if (attr.type == hipMemoryTypeHost && attr.devicePointer) return static_cast<char*>(attr.devicePointer);
  return nullptr;
}

}  // namespace pa

extern "C" int pa_reshape_and_cache(const void* key, const void* value, void* key_cache, void* value_cache,
                                    const int32_t* slot_mapping, int num_tokens, int num_heads, int head_size,
                                    int block_size, int64_t key_stride, int64_t value_stride, int dtype,
                                    void* stream) {
  return pa::token_cache_op<false>("pa_reshape_and_cache", const_cast<void*>(key), const_cast<void*>(value),
                                   key_cache, value_cache, slot_mapping, num_tokens, num_heads, head_size,
                                   block_size, key_stride, value_stride, dtype, stream);
}

extern "C" int pa_rotary_embedding_neox_reshape_and_cache(
    const int64_t* positions, void* query, void* key, const void* value, const void* cos_sin_cache, void* key_cache,
    void* value_cache, const int32_t* slot_mapping, int num_tokens, int num_heads, int head_size, int rot_dim,
    int block_size, int64_t query_stride, int64_t key_stride, int64_t value_stride, int dtype, void* stream) {
  using namespace pa;
  clear_error();
  const char* what = "pa_rotary_embedding_neox_reshape_and_cache";
  if (dtype != PA_DTYPE_F32 && dtype != PA_DTYPE_F16 && dtype != PA_DTYPE_BF16)
    return fail(PA_ERR_UNSUPPORTED, "%s: unsupported data type code: %d", what, dtype);
  PA_REQUIRE(num_tokens >= 0 && num_heads >= 0 && head_size >= 0 && rot_dim >= 0 && block_size > 0,
             "%s: negative size or non-positive block size", what);
  PA_REQUIRE(rot_dim % 2 == 0 && rot_dim <= head_size, "%s: rot_dim %d must be even and <= head_size %d", what,
             rot_dim, head_size);
  if (num_tokens == 0 || num_heads == 0 || head_size == 0) return PA_OK;
  PA_REQUIRE(positions && query && key && value && cos_sin_cache && key_cache && value_cache && slot_mapping,
             "%s: null pointer argument", what);
  const int es = elem_size(dtype);
  const int x = 16 / es;
  const bool rows16 = aligned16(query) && aligned16(key) && aligned16(value) && (query_stride * es) % 16 == 0 &&
                      (key_stride * es) % 16 == 0 && (value_stride * es) % 16 == 0 && (head_size * es) % 16 == 0;
  const bool fused = rows16 && rot_dim > 0 && (rot_dim / 2) % x == 0 && aligned16(cos_sin_cache) &&
                     aligned16(key_cache) && aligned16(value_cache) && (block_size * es) % 16 == 0 &&
                     num_tokens >= kAppendMinTokens && num_tokens < PA_CACHE_RUNS_MIN_TOKENS &&
                     (int64_t)num_heads * head_size < (1ll << 31);
  if (!fused) {
    // Shapes the one-pass kernel does not take (a handful of tokens, prompt-sized batches, odd
    // alignments) run as the two operators they stand for: same results.
    int rc = pa_rotary_embedding_neox(positions, query, key, cos_sin_cache, num_tokens, num_heads, head_size, rot_dim,
                                      query_stride, key_stride, dtype, stream);
    if (rc != PA_OK) return rc;
    return pa_reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, num_tokens, num_heads, head_size,
                                block_size, key_stride, value_stride, dtype, stream);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
#define PA_ROTARY_APPEND(T)                                                                                            \
  launch_rotary_append<T>(positions, query, key, value, cos_sin_cache, key_cache, value_cache, slot_mapping, num_tokens, \
                          num_heads, head_size, rot_dim, block_size, query_stride, key_stride, value_stride, st)
  switch (dtype) {
    case PA_DTYPE_F32: PA_ROTARY_APPEND(float); break;
    case PA_DTYPE_F16: PA_ROTARY_APPEND(_Float16); break;
    default: PA_ROTARY_APPEND(__bf16); break;
  }
#undef PA_ROTARY_APPEND
  PA_CHECK_LAUNCH(what);
  return PA_OK;
}

extern "C" int pa_gather_cached_kv(void* key, void* value, const void* key_cache, const void* value_cache,
                                   const int32_t* slot_mapping, int num_tokens, int num_heads, int head_size,
                                   int block_size, int64_t key_stride, int64_t value_stride, int dtype,
                                   void* stream) {
  return pa::token_cache_op<true>("pa_gather_cached_kv", key, value, const_cast<void*>(key_cache),
                                  const_cast<void*>(value_cache), slot_mapping, num_tokens, num_heads, head_size,
                                  block_size, key_stride, value_stride, dtype, stream);
}

extern "C" int pa_copy_blocks(void* const* key_cache_ptrs, void* const* value_cache_ptrs, int num_layers,
                              const int64_t* block_pairs, int num_pairs, int64_t block_bytes, void* stream) {
  using namespace pa;
  clear_error();
  PA_REQUIRE(num_layers >= 0 && num_pairs >= 0 && block_bytes >= 0, "pa_copy_blocks: negative size");
  if (num_layers == 0 || num_pairs == 0 || block_bytes == 0) return PA_OK;
  PA_REQUIRE(key_cache_ptrs && value_cache_ptrs && block_pairs, "pa_copy_blocks: null pointer argument");
  // widest word that divides the block size and every base pointer
  uintptr_t align_bits = static_cast<uintptr_t>(block_bytes);
  for (int l = 0; l < num_layers; l++) {
    PA_REQUIRE(key_cache_ptrs[l] && value_cache_ptrs[l], "pa_copy_blocks: null cache pointer for layer %d", l);
    align_bits |= reinterpret_cast<uintptr_t>(key_cache_ptrs[l]) | reinterpret_cast<uintptr_t>(value_cache_ptrs[l]);
  }
  for (int p = 0; p < 2 * num_pairs; p++)
    PA_REQUIRE(block_pairs[p] >= 0 && block_pairs[p] < (1ll << 31), "pa_copy_blocks: block number out of range");
  int w = 16;
  while (w > 1 && (align_bits & (uintptr_t)(w - 1))) w >>= 1;
  const int64_t nwords = block_bytes / w;
  // ~16 KiB of 16-byte words per workgroup
  int splits = (int)((nwords + 1023) / 1024);
  if (splits < 1) splits = 1;
  if (splits > 1024) splits = 1024;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CopyTable tab;
  for (int l0 = 0; l0 < num_layers; l0 += kCopyMaxLayers) {
    const int nl = num_layers - l0 < kCopyMaxLayers ? num_layers - l0 : kCopyMaxLayers;
    for (int l = 0; l < nl; l++) {
      tab.key_ptr[l] = static_cast<char*>(key_cache_ptrs[l0 + l]);
      tab.value_ptr[l] = static_cast<char*>(value_cache_ptrs[l0 + l]);
    }
    for (int p0 = 0; p0 < num_pairs; p0 += kCopyMaxPairs) {
      const int np = num_pairs - p0 < kCopyMaxPairs ? num_pairs - p0 : kCopyMaxPairs;
      for (int p = 0; p < np; p++) {
        tab.src[p] = (int32_t)block_pairs[2 * (p0 + p)];
        tab.dst[p] = (int32_t)block_pairs[2 * (p0 + p) + 1];
      }
      const dim3 grid(np, nl, 2 * splits);
      switch (w) {
        case 16: hipLaunchKernelGGL((pa_copy_blocks_kernel<16>), grid, dim3(256), 0, st, tab, block_bytes, splits); break;
        case 8: hipLaunchKernelGGL((pa_copy_blocks_kernel<8>), grid, dim3(256), 0, st, tab, block_bytes, splits); break;
        case 4: hipLaunchKernelGGL((pa_copy_blocks_kernel<4>), grid, dim3(256), 0, st, tab, block_bytes, splits); break;
        case 2: hipLaunchKernelGGL((pa_copy_blocks_kernel<2>), grid, dim3(256), 0, st, tab, block_bytes, splits); break;
        default: hipLaunchKernelGGL((pa_copy_blocks_kernel<1>), grid, dim3(256), 0, st, tab, block_bytes, splits); break;
      }
      PA_CHECK_LAUNCH("pa_copy_blocks launch");
    }
  }
  return PA_OK;
}

extern "C" int pa_swap_blocks(const void* src, void* dst, const int64_t* block_pairs, int num_pairs,
                              int64_t block_bytes, int kind, void* stream) {
  using namespace pa;
  clear_error();
  PA_REQUIRE(num_pairs >= 0 && block_bytes >= 0, "pa_swap_blocks: negative size");
  if (num_pairs == 0 || block_bytes == 0) return PA_OK;
  PA_REQUIRE(src && dst && block_pairs, "pa_swap_blocks: null pointer argument");
  hipMemcpyKind k;
  switch (kind) {
    case PA_COPY_DEVICE_TO_DEVICE: k = hipMemcpyDeviceToDevice; break;
    case PA_COPY_DEVICE_TO_HOST: k = hipMemcpyDeviceToHost; break;
    case PA_COPY_HOST_TO_DEVICE: k = hipMemcpyHostToDevice; break;
    default: return fail(PA_ERR_INVALID_ARGUMENT, "pa_swap_blocks: invalid device combination (kind %d)", kind);
  }
  for (int p = 0; p < 2 * num_pairs; p++)
    PA_REQUIRE(block_pairs[p] >= 0, "pa_swap_blocks: negative block number");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const char* s = static_cast<const char*>(src);
  char* d = static_cast<char*>(dst);
  // Many scattered blocks: one kernel launch per 256 pairs when both sides can be
  // addressed by the device and the blocks are whole 16-byte words.
  static const int kernel_min_pairs = tuning_int("PA_SWAP_KERNEL_MIN_PAIRS", 2);  // tuning builds: 0 = never use the kernel; already at 4 blocks of 128 KiB the kernel moves 27 GB/s against 10
  if (kernel_min_pairs > 0 && num_pairs >= kernel_min_pairs && block_bytes % 16 == 0) {
    char* ds = device_alias(src);
    char* dd = device_alias(dst);
    if (ds && dd && aligned16(ds) && aligned16(dd)) {
      SwapTable tab;
      tab.src = ds;
      tab.dst = dd;
      int splits = (int)((block_bytes / 16 + 4095) / 4096);  // ~64 KiB per workgroup
      if (splits < 1) splits = 1;
      if (splits > 65535) splits = 65535;
      for (int p0 = 0; p0 < num_pairs; p0 += kCopyMaxPairs) {
        const int np = num_pairs - p0 < kCopyMaxPairs ? num_pairs - p0 : kCopyMaxPairs;
        for (int p = 0; p < np; p++) {
          tab.src_block[p] = block_pairs[2 * (p0 + p)];
          tab.dst_block[p] = block_pairs[2 * (p0 + p) + 1];
        }
        hipLaunchKernelGGL(pa_swap_blocks_kernel, dim3(np, splits), dim3(256), 0, st, tab, block_bytes, splits);
        PA_CHECK_LAUNCH("pa_swap_blocks kernel launch");
      }
      return PA_OK;
    }
  }
  // merge runs of consecutive source AND destination blocks into one copy
  int p = 0;
  while (p < num_pairs) {
    int run = 1;
    while (p + run < num_pairs && block_pairs[2 * (p + run)] == block_pairs[2 * p] + run &&
           block_pairs[2 * (p + run) + 1] == block_pairs[2 * p + 1] + run)
      run++;
    hipError_t e = hipMemcpyAsync(d + block_pairs[2 * p + 1] * block_bytes, s + block_pairs[2 * p] * block_bytes,
                                  (size_t)(run * block_bytes), k, st);
    if (e != hipSuccess) return fail(PA_ERR_HIP, "pa_swap_blocks: hipMemcpyAsync: %s", hipGetErrorString(e));
    p += run;
  }
  return PA_OK;
}
