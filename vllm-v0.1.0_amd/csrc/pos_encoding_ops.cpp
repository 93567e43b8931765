// Python module `pos_encoding_ops`: same name, function and argument order as
// vLLM v0.1.0's csrc/pos_encoding.cpp:3-15, so
// vllm/model_executor/layers/attention.py:200-206 calls it unchanged.
#include "torch_shim.h"

namespace {

void rotary_embedding_neox(torch::Tensor& positions,      // [num_tokens]
                           torch::Tensor& query,          // [num_tokens, num_heads * head_size]
                           torch::Tensor& key,            // [num_tokens, num_heads * head_size]
                           int head_size,
                           torch::Tensor& cos_sin_cache)  // [max_position, rot_dim]
{
  using namespace pa_shim;
  const char* what = "rotary_embedding_neox";
  TORCH_CHECK(positions.is_cuda() && query.is_cuda() && key.is_cuda() && cos_sin_cache.is_cuda(), what,
              ": all tensors must be on the GPU");
  const int dtype = dtype_code(query, what);
  TORCH_CHECK(key.scalar_type() == query.scalar_type() && cos_sin_cache.scalar_type() == query.scalar_type(), what,
              ": query, key and cos_sin_cache must share one data type");
  TORCH_CHECK(positions.scalar_type() == at::ScalarType::Long && positions.is_contiguous(), what,
              ": positions must be a contiguous int64 tensor");
  TORCH_CHECK(query.dim() == 2 && key.dim() == 2 && query.sizes() == key.sizes(), what,
              ": query and key must be [num_tokens, num_heads * head_size]");
  TORCH_CHECK(query.size(0) == 0 || (query.stride(1) == 1 && key.stride(1) == 1), what,
              ": query and key rows must be dense");
  TORCH_CHECK(head_size > 0 && query.size(1) % head_size == 0, what, ": hidden size is not a multiple of head_size");
  TORCH_CHECK(cos_sin_cache.dim() == 2 && cos_sin_cache.is_contiguous(), what,
              ": cos_sin_cache must be a contiguous [max_position, rot_dim] tensor");
  const int num_tokens = query.size(0);
  TORCH_CHECK(positions.numel() >= num_tokens, what, ": positions is shorter than the number of tokens");
  // the reference requires equal row strides (pos_encoding_kernels.cu:62)
  TORCH_CHECK(num_tokens == 0 || query.stride(0) == key.stride(0), what, ": query and key row strides differ");
  PA_DEVICE_GUARD(query);
  check_status(pa_rotary_embedding_neox(positions.data_ptr<int64_t>(), query.data_ptr(), key.data_ptr(),
                                        cos_sin_cache.data_ptr(), num_tokens,
                                        static_cast<int>(query.size(1) / head_size), head_size,
                                        static_cast<int>(cos_sin_cache.size(1)),
                                        num_tokens > 0 ? query.stride(0) : 0, num_tokens > 0 ? key.stride(0) : 0,
                                        dtype, current_stream(query)),
               what);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("rotary_embedding_neox", &rotary_embedding_neox,
        "Apply GPT-NeoX style rotary embedding to query and key, in place");
}
