// Python module `attention_ops`: same name, function and argument order as
// vLLM v0.1.0's csrc/attention.cpp:3-19, so
// vllm/model_executor/layers/attention.py:81-91 calls it unchanged.
#include "torch_shim.h"

namespace {

struct DecodeShape {
  int num_seqs, num_heads, head_size, dtype;
};

DecodeShape check_decode_args(const char* what, torch::Tensor& out, torch::Tensor& query, torch::Tensor& key_cache,
                              torch::Tensor& value_cache, torch::Tensor& block_tables, torch::Tensor& context_lens,
                              int block_size) {
  using namespace pa_shim;
  TORCH_CHECK(query.is_cuda() && out.is_cuda() && key_cache.is_cuda() && value_cache.is_cuda() &&
                  block_tables.is_cuda() && context_lens.is_cuda(),
              what, ": all tensors must be on the GPU");
  const int dtype = dtype_code(query, what);
  TORCH_CHECK(out.scalar_type() == query.scalar_type() && key_cache.scalar_type() == query.scalar_type() &&
                  value_cache.scalar_type() == query.scalar_type(),
              what, ": out, query, key_cache and value_cache must share one data type");
  check_token_rows(query, "query");
  TORCH_CHECK(out.dim() == 3 && out.is_contiguous(), what, ": out must be a contiguous [num_seqs, num_heads, head_size]");
  TORCH_CHECK(out.sizes() == query.sizes(), what, ": out and query shapes differ");
  TORCH_CHECK(key_cache.dim() == 5 && value_cache.dim() == 4 && key_cache.is_contiguous() && value_cache.is_contiguous(),
              what, ": key_cache must be a contiguous 5-d tensor and value_cache a contiguous 4-d tensor");
  TORCH_CHECK(block_tables.scalar_type() == at::ScalarType::Int && context_lens.scalar_type() == at::ScalarType::Int,
              what, ": block_tables and context_lens must be int32");
  TORCH_CHECK(block_tables.dim() == 2 && block_tables.is_contiguous() && context_lens.is_contiguous(), what,
              ": block_tables must be a contiguous 2-d tensor and context_lens contiguous");
  const int num_seqs = query.size(0), num_heads = query.size(1), head_size = query.size(2);
  TORCH_CHECK(block_tables.size(0) == num_seqs && context_lens.numel() == num_seqs, what,
              ": block_tables / context_lens do not match the number of sequences");
  TORCH_CHECK(value_cache.size(1) == num_heads && value_cache.size(2) == head_size && value_cache.size(3) == block_size,
              what, ": value_cache shape does not match [num_blocks, ", num_heads, ", ", head_size, ", ", block_size, "]");
  TORCH_CHECK(key_cache.numel() == value_cache.numel() && key_cache.size(3) == block_size, what,
              ": key_cache shape does not match value_cache");
  return DecodeShape{num_seqs, num_heads, head_size, dtype};
}

void single_query_cached_kv_attention(
    torch::Tensor& out,           // [num_seqs, num_heads, head_size]
    torch::Tensor& query,         // [num_seqs, num_heads, head_size]
    torch::Tensor& key_cache,     // [num_blocks, num_heads, head_size/x, block_size, x]
    torch::Tensor& value_cache,   // [num_blocks, num_heads, head_size, block_size]
    float scale,
    torch::Tensor& block_tables,  // [num_seqs, max_num_blocks_per_seq]
    torch::Tensor& context_lens,  // [num_seqs]
    int block_size, int max_context_len) {
  using namespace pa_shim;
  const char* what = "single_query_cached_kv_attention";
  const DecodeShape sh = check_decode_args(what, out, query, key_cache, value_cache, block_tables, context_lens, block_size);
  // The scratch (partial results of split contexts, the launch plan) comes
  // from torch's caching allocator (no hipMalloc in steady state) and is
  // handed back as soon as the launches are queued: the allocator reuses it in
  // stream order, and every launch below is on the current stream.
  const size_t ws_bytes = pa_decode_workspace_bytes(sh.num_seqs, sh.num_heads, sh.head_size, block_size, max_context_len);
  torch::Tensor workspace;
  if (ws_bytes > 0) {
    workspace = torch::empty({static_cast<int64_t>(ws_bytes)}, query.options().dtype(torch::kUInt8));
  }
  PA_DEVICE_GUARD(query);
  check_status(pa_single_query_cached_kv_attention_ws(
                   out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), scale,
                   block_tables.data_ptr<int>(), context_lens.data_ptr<int>(), sh.num_seqs, sh.num_heads, sh.head_size,
                   block_size, static_cast<int>(block_tables.size(1)), max_context_len,
                   sh.num_seqs > 0 ? query.stride(0) : 0, sh.dtype, ws_bytes > 0 ? workspace.data_ptr() : nullptr,
                   ws_bytes, current_stream(query)),
               what);
}

// Not part of the reference's module.  The layers of one decode step all see
// the same context_lens: the caller keeps one workspace for the step (sized by
// decode_workspace_bytes, a uint8 tensor on the GPU), the first layer passes
// plan_ready = false and the launch plan is made into it, the others pass
// true and reuse it.  An empty workspace (size 0) is allowed when none is needed.
// total_context_len: the sum of context_lens if known on the host, else -1.
int64_t decode_workspace_bytes(int num_seqs, int num_heads, int head_size, int block_size, int max_context_len) {
  return static_cast<int64_t>(pa_decode_workspace_bytes(num_seqs, num_heads, head_size, block_size, max_context_len));
}

void single_query_cached_kv_attention_planned(
    torch::Tensor& out, torch::Tensor& query, torch::Tensor& key_cache, torch::Tensor& value_cache, float scale,
    torch::Tensor& block_tables, torch::Tensor& context_lens, int block_size, int max_context_len,
    torch::Tensor& workspace, bool plan_ready, int64_t total_context_len) {
  using namespace pa_shim;
  const char* what = "single_query_cached_kv_attention_planned";
  const DecodeShape sh = check_decode_args(what, out, query, key_cache, value_cache, block_tables, context_lens, block_size);
  TORCH_CHECK(workspace.numel() == 0 || (workspace.is_cuda() && workspace.is_contiguous() &&
                                         workspace.scalar_type() == at::ScalarType::Byte),
              what, ": workspace must be a contiguous uint8 tensor on the GPU");
  const size_t ws_bytes = static_cast<size_t>(workspace.numel());
  PA_DEVICE_GUARD(query);
  check_status(pa_single_query_cached_kv_attention_planned(
                   out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), scale,
                   block_tables.data_ptr<int>(), context_lens.data_ptr<int>(), sh.num_seqs, sh.num_heads, sh.head_size,
                   block_size, static_cast<int>(block_tables.size(1)), max_context_len,
                   sh.num_seqs > 0 ? query.stride(0) : 0, sh.dtype, ws_bytes > 0 ? workspace.data_ptr() : nullptr,
                   ws_bytes, plan_ready ? 1 : 0, total_context_len, current_stream(query)),
               what);
}

// Not part of the reference's module.  The generation branch of PagedAttention.forward
// (vllm/model_executor/layers/attention.py:134-161) in one launch: reshape_and_cache of the
// step's new token of every sequence, then single_query_cached_kv_attention over the context
// that now ends with it.  The new token's place is not taken from a slot_mapping: it is the
// last position of the context, block_tables[seq][(context_lens[seq] - 1) / block_size] at offset
// (context_lens[seq] - 1) % block_size, which is where the reference's worker puts a generation
// token (vllm/worker/worker.py:183-190).  key / value: [num_seqs, num_heads, head_size] token rows.
void reshape_and_cache_single_query_cached_kv_attention(
    torch::Tensor& out, torch::Tensor& query, torch::Tensor& key, torch::Tensor& value, torch::Tensor& key_cache,
    torch::Tensor& value_cache, float scale, torch::Tensor& block_tables, torch::Tensor& context_lens, int block_size,
    int max_context_len, torch::Tensor& workspace, bool plan_ready, int64_t total_context_len) {
  using namespace pa_shim;
  const char* what = "reshape_and_cache_single_query_cached_kv_attention";
  const DecodeShape sh = check_decode_args(what, out, query, key_cache, value_cache, block_tables, context_lens, block_size);
  TORCH_CHECK(key.is_cuda() && value.is_cuda() && key.scalar_type() == query.scalar_type() &&
                  value.scalar_type() == query.scalar_type(),
              what, ": key and value must be on the GPU and of query's data type");
  check_token_rows(key, "key");
  check_token_rows(value, "value");
  TORCH_CHECK(key.sizes() == query.sizes() && value.sizes() == query.sizes(), what, ": query, key and value shapes differ");
  TORCH_CHECK(workspace.numel() == 0 || (workspace.is_cuda() && workspace.is_contiguous() &&
                                         workspace.scalar_type() == at::ScalarType::Byte),
              what, ": workspace must be a contiguous uint8 tensor on the GPU");
  const size_t ws_bytes = static_cast<size_t>(workspace.numel());
  PA_DEVICE_GUARD(query);
  check_status(pa_append_single_query_cached_kv_attention(
                   out.data_ptr(), query.data_ptr(), key.data_ptr(), value.data_ptr(), key_cache.data_ptr(),
                   value_cache.data_ptr(), scale, block_tables.data_ptr<int>(), context_lens.data_ptr<int>(), sh.num_seqs,
                   sh.num_heads, sh.head_size, block_size, static_cast<int>(block_tables.size(1)), max_context_len,
                   sh.num_seqs > 0 ? query.stride(0) : 0, sh.num_seqs > 0 ? key.stride(0) : 0,
                   sh.num_seqs > 0 ? value.stride(0) : 0, sh.dtype, ws_bytes > 0 ? workspace.data_ptr() : nullptr, ws_bytes,
                   plan_ready ? 1 : 0, total_context_len, current_stream(query)),
               what);
}

// Not part of the reference's module (its prompt attention lives in xformers):
// the causal per-prompt attention PagedAttention.multi_query_kv_attention needs
// (vllm/model_executor/layers/attention.py:44-70), with the prompt lengths as
// the host list BlockDiagonalCausalMask.from_seqlens takes.
void multi_query_kv_attention(
    torch::Tensor& out,     // [num_prompt_tokens, num_heads, head_size]
    torch::Tensor& query,   // [num_prompt_tokens, num_heads, head_size]
    torch::Tensor& key, torch::Tensor& value, const std::vector<int64_t>& seq_lens, float scale) {
  using namespace pa_shim;
  const char* what = "multi_query_kv_attention";
  TORCH_CHECK(out.is_cuda() && query.is_cuda() && key.is_cuda() && value.is_cuda(), what, ": all tensors must be on the GPU");
  const int dtype = dtype_code(query, what);
  TORCH_CHECK(out.scalar_type() == query.scalar_type() && key.scalar_type() == query.scalar_type() &&
                  value.scalar_type() == query.scalar_type(),
              what, ": out, query, key and value must share one data type");
  check_token_rows(query, "query");
  check_token_rows(key, "key");
  check_token_rows(value, "value");
  TORCH_CHECK(out.dim() == 3 && out.is_contiguous() && out.sizes() == query.sizes(), what,
              ": out must be contiguous and shaped like query");
  TORCH_CHECK(key.sizes() == query.sizes() && value.sizes() == query.sizes(), what, ": query, key and value shapes differ");
  std::vector<int32_t> cu(seq_lens.size() + 1, 0);
  for (size_t i = 0; i < seq_lens.size(); ++i) {
    TORCH_CHECK(seq_lens[i] >= 0, what, ": negative prompt length");
    TORCH_CHECK(cu[i] + seq_lens[i] <= query.size(0), what, ": prompt lengths exceed the number of token rows");
    cu[i + 1] = cu[i] + static_cast<int32_t>(seq_lens[i]);
  }
  const int64_t row = query.size(1) * query.size(2);
  const bool any = query.size(0) > 0;
  PA_DEVICE_GUARD(query);
  check_status(pa_multi_query_kv_attention(out.data_ptr(), query.data_ptr(), key.data_ptr(), value.data_ptr(), cu.data(),
                                           static_cast<int>(seq_lens.size()), static_cast<int>(query.size(1)),
                                           static_cast<int>(query.size(2)), scale, any ? query.stride(0) : row,
                                           any ? key.stride(0) : row, any ? value.stride(0) : row, dtype,
                                           current_stream(query)),
               what);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("single_query_cached_kv_attention", &single_query_cached_kv_attention,
        "Attention of one new query token per sequence over the paged key/value cache (MI355X HIP kernel)");
  m.def("decode_workspace_bytes", &decode_workspace_bytes,
        "Bytes of workspace single_query_cached_kv_attention_planned wants for these sizes (0 = none)");
  m.def("single_query_cached_kv_attention_planned", &single_query_cached_kv_attention_planned,
        "single_query_cached_kv_attention with a workspace whose launch plan is shared by the layers of a step");
  m.def("reshape_and_cache_single_query_cached_kv_attention", &reshape_and_cache_single_query_cached_kv_attention,
        "reshape_and_cache of every sequence's new token and single_query_cached_kv_attention over the context "
        "that ends with it, in one launch");
  m.def("multi_query_kv_attention", &multi_query_kv_attention,
        "Causal self-attention inside each prompt of a flat token batch (MI355X matrix-core kernel)");
}
