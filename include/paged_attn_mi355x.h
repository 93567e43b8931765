/*
 * paged_attn_mi355x — C ABI of the MI355X (gfx950) PagedAttention decode path.
 *
 * One shared library, libpaged_attn_mi355x.so, replaces the device side of
 * vLLM v0.1.0's two operator modules:
 *
 *   vllm.attention_ops   csrc/attention.cpp:3-19
 *   vllm.cache_ops       csrc/cache.cpp:6-47
 *
 * Every entry point takes plain pointers and sizes (no torch types), launches
 * its work asynchronously on the HIP stream it is given (NULL = the default
 * stream) and returns PA_OK or an error code; pa_last_error() then holds a
 * message for the calling thread.  Nothing is allocated, freed or synchronised
 * by any entry point, so all of them are safe under HIP graph capture, except
 * that pa_swap_blocks issues hipMemcpyAsync calls (capturable, but they become
 * memcpy nodes).  All device pointers must belong to the current HIP device.
 *
 * Element types: see pa_dtype.  Cache layouts, with x = 16 / sizeof(element)
 * (vllm/worker/cache_engine.py:48-63):
 *
 *   key_cache    [num_blocks, num_heads, head_size / x, block_size, x]
 *   value_cache  [num_blocks, num_heads, head_size, block_size]
 *
 * All offsets are computed in 64 bits: a cache tensor may hold more than 2^31
 * elements (the reference's kernels index with 32-bit ints).
 */
#ifndef PAGED_ATTN_MI355X_H_
#define PAGED_ATTN_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PA_ABI_VERSION 1

typedef enum pa_dtype {
  PA_DTYPE_F32 = 0,  /* float                       (at::ScalarType::Float)    */
  PA_DTYPE_F16 = 1,  /* IEEE binary16               (at::ScalarType::Half)     */
  PA_DTYPE_BF16 = 2  /* bfloat16                    (at::ScalarType::BFloat16) */
} pa_dtype;

typedef enum pa_status {
  PA_OK = 0,
  PA_ERR_INVALID_ARGUMENT = 1, /* null pointer, negative size, misalignment   */
  PA_ERR_UNSUPPORTED = 2,      /* dtype / head_size / block_size not built    */
  PA_ERR_HIP = 3               /* a HIP runtime call or kernel launch failed  */
} pa_status;

typedef enum pa_copy_kind {
  PA_COPY_DEVICE_TO_DEVICE = 0,
  PA_COPY_DEVICE_TO_HOST = 1,
  PA_COPY_HOST_TO_DEVICE = 2
} pa_copy_kind;

/* ABI version of the loaded library (== PA_ABI_VERSION it was built with). */
int pa_abi_version(void);

/* Message of the last failing call on this thread ("" if none). */
const char* pa_last_error(void);

/*
 * Decode attention of one new token per sequence over the paged KV cache.
 * Replaces single_query_cached_kv_attention
 * (csrc/attention/attention_kernels.cu:450-469; kernel :67-314).
 *
 *   out          [num_seqs, num_heads, head_size], dense
 *   query        [num_seqs, num_heads, head_size], row stride q_stride elements
 *                (q is a column slice of the fused qkv projection)
 *   block_tables [num_seqs, max_num_blocks_per_seq] int32; entries past
 *                ceil(context_len / block_size) are never read
 *   context_lens [num_seqs] int32, each >= 0 (0 writes zeros for that row)
 *
 * out = softmax(scale * q . K^T) . V per (sequence, head), with the softmax
 * denominator (sum + 1e-6) of the reference kernel (:213).  Supported:
 * head_size in {64, 80, 96, 128}, block_size in {8, 16, 32}, all pa_dtype
 * values — the set the reference instantiates (:370-400, :416-448); anything
 * else returns PA_ERR_UNSUPPORTED, where the reference raises.
 * max_context_len is accepted for signature parity and only validated: the
 * kernel keeps no per-context buffer, so context length is not limited by LDS.
 */
int pa_single_query_cached_kv_attention(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    float scale, const int32_t* block_tables, const int32_t* context_lens,
    int num_seqs, int num_heads, int head_size, int block_size,
    int max_num_blocks_per_seq, int max_context_len, int64_t q_stride,
    int dtype, void* stream);

/*
 * Same operation with a caller-provided scratch buffer.  With a workspace of
 * at least pa_decode_workspace_bytes(...) bytes (device memory, 8-byte
 * aligned, contents irrelevant, may be reused by the next call on the same
 * stream) the library may split every sequence's context into partitions that
 * run as separate workgroups and are combined by a second kernel; it does so
 * when num_seqs * num_heads alone cannot fill the GPU (small batches), and for
 * big batches whose longest context exceeds one partition of 256 to 1024 tokens
 * (the more rows, the longer; there, the partitions
 * that exist are listed on the device first, so that one long sequence among
 * short ones does not set the launch time alone).  With
 * workspace == NULL, or one that is too small, it behaves exactly like
 * pa_single_query_cached_kv_attention.  Results agree with the unsplit kernel
 * up to fp32 summation order.  No counterpart in the reference, whose kernel
 * always runs one thread block per (sequence, head) (:367).
 *
 * pa_decode_workspace_bytes returns 0 when the library has no use for a
 * workspace.
 */
size_t pa_decode_workspace_bytes(int num_seqs, int num_heads, int head_size,
                                 int block_size, int max_context_len);

int pa_single_query_cached_kv_attention_ws(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    float scale, const int32_t* block_tables, const int32_t* context_lens,
    int num_seqs, int num_heads, int head_size, int block_size,
    int max_num_blocks_per_seq, int max_context_len, int64_t q_stride,
    int dtype, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Same again, for callers that run the operation several times on the same
 * context_lens — every layer of a decode step does.  The head of the
 * workspace holds a plan that depends on context_lens only: for batches of 64
 * to 1024 sequences the order in which the (sequence, head) rows are dealt to
 * the GPU's eight XCDs (sequences placed by length, so that the XCDs finish
 * together: worth 4 % on a batch of random lengths, 13 % on long-tailed
 * lengths), and for big batches with long contexts the list of partitions.
 * The first call of a step passes plan_ready = 0 and makes the plan; the
 * others pass plan_ready = 1 with the SAME workspace, num_seqs, num_heads,
 * head_size, block_size, max_context_len and an unchanged context_lens, and
 * skip that kernel.  The rest of the workspace is scratch as above.  The plan
 * carries a stamp of the sizes it was made for: with a workspace whose plan
 * was never made, or made for other sizes, the call fills `out` with NaN
 * instead of computing anything, so that the caller's error cannot pass for a
 * result.  A plan reused after context_lens changed (same sizes, other
 * contents) is caught per sequence: the plan records the length of every
 * sequence it placed or partitioned, and the output rows of a sequence whose
 * length differs from the recorded one are filled with NaN (the other rows
 * are correct: the plan only decides where and in how many pieces a row is
 * computed).  pa_single_query_cached_kv_attention_ws makes the
 * plan on every call, and makes the row order only for launches big enough to
 * win it back.  total_context_len is the sum of context_lens if the caller
 * knows it on the host (the engine builds context_lens from a host list), else
 * -1: with it the library can tell on the host that no sequence of a big batch
 * is long enough to need partitions, and leaves that machinery out of the
 * launch, and reserves workgroup ids only for as many partitions as that sum
 * allows; it can also tell that all sequences are (nearly) as long as the
 * longest, and then gives a mid-size batch of long contexts two uniform
 * partitions per row instead of the item list.  It must be the same for all
 * calls on one plan, and it must not be
 * less than the true sum: if the device finds more partitions than were
 * reserved, `out` is filled with NaN.  No counterpart in the reference.
 */
int pa_single_query_cached_kv_attention_planned(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    float scale, const int32_t* block_tables, const int32_t* context_lens,
    int num_seqs, int num_heads, int head_size, int block_size,
    int max_num_blocks_per_seq, int max_context_len, int64_t q_stride,
    int dtype, void* workspace, size_t workspace_bytes, int plan_ready,
    int64_t total_context_len, void* stream);

/*
 * The _ws form for callers that cannot allocate device memory: code under HIP
 * graph capture, serving loops and harnesses that forbid hipMalloc.  No
 * counterpart in the reference.  The scratch (partition results, item list,
 * row table) is a fixed array of PA_DECODE_STATIC_SCRATCH_BYTES inside the
 * library's code object, one copy per GPU, held from library load until the
 * process ends; the call allocates and frees nothing.  A launch whose
 * pa_decode_workspace_bytes(...) exceeds that size returns
 * PA_ERR_INVALID_ARGUMENT and launches nothing (256 sequences x 32 heads x
 * head size 128 with contexts up to 4096 tokens need 17 MiB; use the _ws form
 * beyond).  Results are those of the _ws form on the same arguments.  All
 * calls on one GPU share that GPU's array: calls that overlap in time on one
 * GPU (different streams, or graphs replayed concurrently) race on it; calls on
 * one stream are ordered and safe.  Nothing in the array has to survive from
 * one call to the next or start out as zero: every call rewrites what it
 * reads.  The first call on a device looks the array's address up
 * (hipGetSymbolAddress); make that call outside graph capture.
 */
#define PA_DECODE_STATIC_SCRATCH_BYTES ((size_t)32 << 20)

int pa_single_query_cached_kv_attention_static(
    void* out, const void* query, const void* key_cache, const void* value_cache,
    float scale, const int32_t* block_tables, const int32_t* context_lens,
    int num_seqs, int num_heads, int head_size, int block_size,
    int max_num_blocks_per_seq, int max_context_len, int64_t q_stride,
    int dtype, void* stream);

/*
 * A decode step's two cache-side operations in one launch: reshape_and_cache
 * of every sequence's new token, then single_query_cached_kv_attention over
 * the context that now ends with it.  Replaces the generation branch of
 * PagedAttention.forward (vllm/model_executor/layers/attention.py:134-161),
 * which runs the two kernels back to back (csrc/cache_kernels.cu:186-224,
 * csrc/attention/attention_kernels.cu:332-401).
 *
 *   key, value   [num_seqs, num_heads, head_size], row strides in elements:
 *                the new token of each sequence
 *   context_lens count the new token.  Its place is the last position of the
 *                context: block block_tables[seq][(context_lens[seq] - 1) /
 *                block_size], offset (context_lens[seq] - 1) % block_size —
 *                the slot the reference's worker computes for a generation
 *                token (vllm/worker/worker.py:183-190); there is no
 *                slot_mapping argument.
 *
 * The workgroup that reads a sequence's last block puts the new key and value
 * in place in registers, uses them, and writes them to the caches (the key as
 * 16-byte vectors, the value tile whole), which saves the separate launch and
 * its read-modify-write of every last block.  That pays for SMALL launches
 * only.  Measured on MI355X per layer, 32 heads x 128, 1024 tokens, against
 * the two calls: 1 sequence 18.4 -> 11.5 us, 4: 21.3 -> 19.6, 16: 48.1 -> 45.5,
 * but 32: 93.7 -> 96.8, 128: 333 -> 399, 256: 657 -> 731 — once the launch
 * saturates HBM with reads, stores trickling in all through it cost more than
 * the same stores in a launch of their own.  Callers should use it up to a
 * few hundred (sequence, head) rows and keep the two calls beyond.
 * `out` and both caches end up bit-identical to pa_reshape_and_cache followed
 * by pa_single_query_cached_kv_attention_planned on the same arguments.  The
 * last block of a sequence must not be mapped by another sequence of the batch
 * (the reference's block manager copies a shared last block before appending:
 * vllm/core/block_manager.py:139-160), and nothing else may write the caches
 * while the call is in flight.  A sequence with context_lens[seq] == 0, or
 * whose context exceeds the table's width, gets nothing written.
 * workspace / workspace_bytes / total_context_len as for the _planned form;
 * workspace may be NULL (no partitions, rows in batch order).  plan_ready:
 * 0 = make the plan, to be shared by later calls; 1 = the plan in the
 * workspace is ready; -1 = make a plan for this call alone (as the _ws form).
 */
int pa_append_single_query_cached_kv_attention(
    void* out, const void* query, const void* key, const void* value,
    void* key_cache, void* value_cache, float scale,
    const int32_t* block_tables, const int32_t* context_lens, int num_seqs,
    int num_heads, int head_size, int block_size, int max_num_blocks_per_seq,
    int max_context_len, int64_t q_stride, int64_t key_stride,
    int64_t value_stride, int dtype, void* workspace, size_t workspace_bytes,
    int plan_ready, int64_t total_context_len, void* stream);

/*
 * Scatter new key/value vectors into the paged caches.
 * Replaces reshape_and_cache (csrc/cache_kernels.cu:186-224; kernel :143-182).
 *
 *   key, value   [num_tokens, num_heads, head_size], row strides in elements
 *   slot_mapping [num_tokens] int32; slot = block * block_size + offset.
 *                A negative slot skips the token (the reference has no such
 *                case; this keeps padded batches from writing out of bounds).
 *
 * For batches of 16 to 511 tokens a token that is alone in its block (every
 * token of a decode step is) has the block's whole value tile read and written
 * back with the new elements in place: the other slots of that block keep
 * their bytes, but nothing else may write to the block while the call is in
 * flight (the reference orders swaps and copies before the model's kernels:
 * vllm/model_executor/layers/attention.py:125-126, worker/cache_engine.py).
 */
int pa_reshape_and_cache(
    const void* key, const void* value, void* key_cache, void* value_cache,
    const int32_t* slot_mapping, int num_tokens, int num_heads, int head_size,
    int block_size, int64_t key_stride, int64_t value_stride, int dtype,
    void* stream);

/*
 * Inverse of pa_reshape_and_cache: paged caches -> [num_tokens, H, D] rows.
 * Replaces gather_cached_kv (csrc/cache_kernels.cu:348-386).
 */
int pa_gather_cached_kv(
    void* key, void* value, const void* key_cache, const void* value_cache,
    const int32_t* slot_mapping, int num_tokens, int num_heads, int head_size,
    int block_size, int64_t key_stride, int64_t value_stride, int dtype,
    void* stream);

/*
 * For every layer and every (src, dst) pair copy one key block and one value
 * block inside the same cache tensor.  Replaces copy_blocks
 * (csrc/cache_kernels.cu:82-138; kernel :53-78).
 *
 *   key_cache_ptrs, value_cache_ptrs  HOST arrays of num_layers device pointers
 *   block_pairs                       HOST array of 2 * num_pairs block numbers
 *                                     (src0, dst0, src1, dst1, ...)
 *   block_bytes                       bytes of one block of one cache tensor
 *
 * The pointer and pair tables travel in the kernel arguments, so unlike the
 * reference (three blocking host-to-device copies, :116-121) the call neither
 * allocates nor synchronises.  The host arrays may be reused as soon as the
 * call returns.  Pairs are copied concurrently: a block must not be both a
 * source and a destination in one call.
 */
int pa_copy_blocks(
    void* const* key_cache_ptrs, void* const* value_cache_ptrs, int num_layers,
    const int64_t* block_pairs, int num_pairs, int64_t block_bytes,
    void* stream);

/*
 * Whole-block copies between two different cache tensors (GPU<->GPU on one
 * device, GPU->pinned host, pinned host->GPU).  Replaces swap_blocks
 * (csrc/cache_kernels.cu:9-47).  block_pairs as in pa_copy_blocks (HOST).
 * Two or more pairs are moved by one kernel launch per 256 pairs when both
 * sides are addressable by the device (device memory, or pinned host memory,
 * which is mapped): scattered 128 KiB blocks then move at PCIe rate instead of
 * at the rate the host can issue copies.  Otherwise runs of consecutive
 * (src, dst) block numbers are merged into one hipMemcpyAsync each.
 */
int pa_swap_blocks(
    const void* src, void* dst, const int64_t* block_pairs, int num_pairs,
    int64_t block_bytes, int kind, void* stream);

/* ---- prompt attention ------------------------------------------------------- */

/*
 * Causal self-attention inside each prompt of a flat token batch:
 *   out[i] = sum_{j <= i, same prompt} softmax_j(scale * <q_i, k_j>) v_j
 * Replaces the call PagedAttention.multi_query_kv_attention makes into
 * xformers' memory_efficient_attention_forward with a BlockDiagonalCausalMask
 * built from the prompt lengths (vllm/model_executor/layers/attention.py:44-70;
 * restated per prompt in tests/kernels/test_attention.py:69-101).
 *
 *   out         [num_tokens, num_heads, head_size] dense
 *   query, key, value
 *               [num_tokens, num_heads, head_size], row strides in elements
 *               (the q, k, v column slices of a fused qkv projection)
 *   cu_seqlens  HOST array of num_seqs + 1 non-decreasing token offsets: prompt
 *               s owns token rows cu_seqlens[s] .. cu_seqlens[s + 1] - 1 (the
 *               prompt lengths are host data in the reference too).  Rows outside
 *               every prompt (padding) are neither read nor written.
 *
 * Head sizes 64, 80, 96, 128; fp16 / bf16 on the 16-bit matrix instructions with
 * fp32 accumulation and fp32 softmax, fp32 on the fp32 matrix instructions.
 * Pointers and row strides must be multiples of 16 bytes.  The offsets ride in
 * the kernel arguments: no host-to-device copy, no synchronisation.
 */
int pa_multi_query_kv_attention(
    void* out, const void* query, const void* key, const void* value,
    const int32_t* cu_seqlens, int num_seqs, int num_heads, int head_size,
    float scale, int64_t q_stride, int64_t k_stride, int64_t v_stride,
    int dtype, void* stream);

/* ---- operators next to the attention path in a decoder layer ------------- */

/*
 * GPT-NeoX style rotary embedding, in place on query and key.  Replaces
 * rotary_embedding_neox (csrc/pos_encoding_kernels.cu:51-83; kernel :7-47).
 *
 *   positions      [num_tokens] int64
 *   query, key     [num_tokens, num_heads * head_size], row strides in elements
 *   cos_sin_cache  [max_position, rot_dim] = cos(rot_dim/2) | sin(rot_dim/2) per row
 *
 * Only the first rot_dim elements of each head are rotated (rot_dim <= head_size,
 * even).  Arithmetic is done in the storage type, one rounding per operation, as
 * the reference's kernel does.
 */
int pa_rotary_embedding_neox(
    const int64_t* positions, void* query, void* key, const void* cos_sin_cache,
    int num_tokens, int num_heads, int head_size, int rot_dim,
    int64_t query_stride, int64_t key_stride, int dtype, void* stream);

/*
 * pa_rotary_embedding_neox followed by pa_reshape_and_cache, in one pass where the
 * shapes allow.  Replaces the pair of calls a rotary attention layer makes for the
 * tokens it caches (vllm/model_executor/layers/attention.py:200-206 rotary_embedding_neox,
 * then :134-140 reshape_and_cache; kernels csrc/pos_encoding_kernels.cu:7-47 and
 * csrc/cache_kernels.cu:143-182).  query and key are rotated in place, the rotated
 * key and the value are written to the caches; query, key and both caches end up
 * bit-identical to the two separate calls.  Decode-sized batches (16 to 511 tokens,
 * 16-byte aligned rows, rot_dim / 2 a multiple of 16 bytes) are one kernel that reads
 * the key once; everything else is run as the two operators.  A negative slot
 * rotates the token and caches nothing.
 */
int pa_rotary_embedding_neox_reshape_and_cache(
    const int64_t* positions, void* query, void* key, const void* value,
    const void* cos_sin_cache, void* key_cache, void* value_cache,
    const int32_t* slot_mapping, int num_tokens, int num_heads, int head_size,
    int rot_dim, int block_size, int64_t query_stride, int64_t key_stride,
    int64_t value_stride, int dtype, void* stream);

/*
 * out = ((T)(x * rsqrt(mean(x^2) + epsilon))) * weight per row.  Replaces
 * rms_norm (csrc/layernorm_kernels.cu:38-63; kernel :10-34).  out, input:
 * [num_tokens, hidden_size] dense; weight: [hidden_size].
 */
int pa_rms_norm(
    void* out, const void* input, const void* weight, float epsilon,
    int num_tokens, int hidden_size, int dtype, void* stream);

/*
 * out[t, i] = silu(input[t, i]) * input[t, d + i].  Replaces silu_and_mul
 * (csrc/activation_kernels.cu:27-48; kernel :13-23).  input: [num_tokens, 2 * d]
 * dense; out: [num_tokens, d] dense.
 */
int pa_silu_and_mul(
    void* out, const void* input, int num_tokens, int d, int dtype, void* stream);

/*
 * One-shot all-reduce (sum) for the small messages of tensor parallelism: the
 * partial results of a row-parallel matmul, which the reference sums with
 * NCCL's all-reduce replayed from a CUDA graph
 * (vllm/model_executor/parallel_utils/parallel_state.py:506-556; call site
 * parallel_utils/tensor_parallel/layers.py:416-426; group set-up
 * vllm/worker/worker.py:286-301).  On an MI355X node every pair of GPUs has its
 * own xGMI link: each rank pushes its buffer into an inbox on every peer over
 * all links at once, raises a flag there, waits for its peers' flags and sums
 * the contributions from local memory, in rank order and in fp32, so that
 * all ranks end with bit-identical sums.  One kernel launch per reduce, no
 * host involvement, capturable in a HIP graph (the reduce's sequence number
 * lives in device memory).  Meant for messages up to a few MiB and at most 8
 * ranks of one node; larger messages belong to RCCL.
 *
 * These are the only entry points of the library that allocate, free or
 * synchronise, and only in create / connect / status / destroy:
 *
 *   pa_comm_create   on the current device: control words and an inbox of
 *                    2 x world x max_bytes; writes PA_COMM_HANDLE_BYTES of IPC
 *                    handles to handles_out for peers in other processes.
 *   pa_comm_connect  all_handles = the world's handle blobs in rank order
 *                    (exchanged by the caller, e.g. with an all_gather over
 *                    its process group); maps every peer's memory.
 *   pa_comm_connect_local  the same for communicators that live in this
 *                    process (one per device, or several on one device): takes
 *                    the communicators themselves, in rank order.
 *   Every rank must have connected before any rank's first reduce (the caller
 *   places a barrier of its process group in between).
 *   pa_comm_all_reduce  data[count] of dtype, 16-byte aligned, count x element
 *                    size a multiple of 16 and at most max_bytes: summed over
 *                    the ranks in place, asynchronously on `stream`.  All ranks
 *                    must issue the same sequence of reduces.  Reduces of one
 *                    communicator must not overlap in time (one stream, or
 *                    event-ordered).  A rank whose peers do not arrive within
 *                    the wait budget gives up the wait, fills its result with
 *                    NaN and counts a timeout; the communicator is broken from
 *                    then on and must be replaced: later reduces on it do not
 *                    wait any more — one whose peers' flags are already there
 *                    still returns the sum, every other one NaN and another
 *                    count (a dead peer costs one bounded wait, not one per
 *                    reduce).  The result is not symmetric: the rank that gave
 *                    up holds NaN while its peers may hold a valid sum.
 *   pa_comm_set_wait_budget  how long a reduce waits for its peers, in seconds
 *                    (default 30; applies to launches made, or captured, after
 *                    the call).  It has to cover the longest stall a healthy
 *                    rank can have between two reduces (graph capture, a code
 *                    object loading, a profiler attaching).
 *   pa_comm_status   synchronous read-back: reduces completed, timeouts.  A
 *                    caller checks it where it synchronises anyway (the end of
 *                    a step) and stops on a timeout rather than use the NaNs.
 */
#define PA_COMM_HANDLE_BYTES 128

/*
 * algorithm: PA_COMM_ONE_SHOT as described above (the whole message to every
 * peer, one round of flags); PA_COMM_TWO_SHOT: rank r owns the r-th part of
 * the message — every rank pushes each peer the part that peer owns, the
 * owners sum and push the sums to everybody, two rounds of flags, 2 / N of the
 * message per link; PA_COMM_AUTO picks two-shot for more than two ranks once
 * it saves at least 256 KiB per link.  All ranks must pass the same value.
 */
typedef enum pa_comm_algorithm {
  PA_COMM_AUTO = 0,
  PA_COMM_ONE_SHOT = 1,
  PA_COMM_TWO_SHOT = 2
} pa_comm_algorithm;

typedef struct pa_comm pa_comm;

int pa_comm_create(int rank, int world, size_t max_bytes, pa_comm** comm,
                   unsigned char* handles_out);
int pa_comm_connect(pa_comm* comm, const unsigned char* all_handles);
int pa_comm_connect_local(pa_comm* comm, pa_comm* const* all_comms);
int pa_comm_all_reduce(pa_comm* comm, void* data, size_t count, int dtype,
                       int algorithm, void* stream);
int pa_comm_set_wait_budget(pa_comm* comm, double seconds);
int pa_comm_status(pa_comm* comm, uint32_t* reduces_done, uint32_t* timeouts);
int pa_comm_destroy(pa_comm* comm);

#ifdef __cplusplus
}
#endif

#endif /* PAGED_ATTN_MI355X_H_ */
