// Python module `cache_ops`: same names, functions and argument order as
// vLLM v0.1.0's csrc/cache.cpp:6-47, so vllm/worker/cache_engine.py:101-130
// and vllm/model_executor/layers/attention.py:134-140 call it unchanged.
#include <map>
#include <vector>

#include "torch_shim.h"

namespace {

using pa_shim::check_status;
using pa_shim::check_token_rows;
using pa_shim::current_stream;
using pa_shim::dtype_code;

// csrc/cache_kernels.cu:9-47
void swap_blocks(torch::Tensor& src, torch::Tensor& dst, const std::map<int64_t, int64_t>& block_mapping) {
  const char* what = "swap_blocks";
  int kind;
  const torch::Tensor* gpu_side;
  if (src.is_cuda() && dst.is_cuda()) {
    TORCH_CHECK(src.device().index() == dst.device().index(), "src and dst must be on the same GPU");
    kind = PA_COPY_DEVICE_TO_DEVICE;
    gpu_side = &src;
  } else if (src.is_cuda() && dst.is_cpu()) {
    kind = PA_COPY_DEVICE_TO_HOST;
    gpu_side = &src;
  } else if (src.is_cpu() && dst.is_cuda()) {
    kind = PA_COPY_HOST_TO_DEVICE;
    gpu_side = &dst;
  } else {
    TORCH_CHECK(false, "Invalid device combination");
  }
  TORCH_CHECK(src.is_contiguous() && dst.is_contiguous(), what, ": src and dst must be contiguous");
  TORCH_CHECK(src.dim() >= 1 && dst.dim() == src.dim() && src.scalar_type() == dst.scalar_type(), what,
              ": src and dst must have the same rank and data type");
  if (block_mapping.empty() || src.size(0) == 0) return;
  const int64_t block_bytes = src.element_size() * src[0].numel();
  TORCH_CHECK(dst.size(0) > 0 && block_bytes == dst.element_size() * dst[0].numel(), what,
              ": src and dst blocks differ in size");
  std::vector<int64_t> pairs;
  pairs.reserve(2 * block_mapping.size());
  for (const auto& kv : block_mapping) {
    TORCH_CHECK(kv.first >= 0 && kv.first < src.size(0) && kv.second >= 0 && kv.second < dst.size(0), what,
                ": block number out of range: ", kv.first, " -> ", kv.second);
    pairs.push_back(kv.first);
    pairs.push_back(kv.second);
  }
  PA_DEVICE_GUARD(*gpu_side);
  check_status(pa_swap_blocks(src.data_ptr(), dst.data_ptr(), pairs.data(), static_cast<int>(block_mapping.size()),
                              block_bytes, kind, current_stream(*gpu_side)),
               what);
}

// csrc/cache_kernels.cu:82-138
void copy_blocks(std::vector<torch::Tensor>& key_caches, std::vector<torch::Tensor>& value_caches,
                 const std::map<int64_t, std::vector<int64_t>>& block_mapping) {
  const char* what = "copy_blocks";
  const int num_layers = static_cast<int>(key_caches.size());
  TORCH_CHECK(num_layers == static_cast<int>(value_caches.size()), what,
              ": key_caches and value_caches differ in length");
  if (num_layers == 0) return;
  const torch::Device device = key_caches[0].device();
  TORCH_CHECK(device.is_cuda(), what, ": caches must be on the GPU");
  const int64_t num_blocks = key_caches[0].size(0);
  if (num_blocks == 0) return;
  const int64_t block_bytes = key_caches[0].element_size() * key_caches[0][0].numel();
  std::vector<void*> kptrs(num_layers), vptrs(num_layers);
  for (int l = 0; l < num_layers; l++) {
    const torch::Tensor& k = key_caches[l];
    const torch::Tensor& v = value_caches[l];
    TORCH_CHECK(k.device() == device && v.device() == device, what, ": all caches must be on one device");
    TORCH_CHECK(k.is_contiguous() && v.is_contiguous(), what, ": caches must be contiguous");
    TORCH_CHECK(k.size(0) == num_blocks && v.size(0) == num_blocks &&
                    k.element_size() * k[0].numel() == block_bytes && v.element_size() * v[0].numel() == block_bytes,
                what, ": all caches must have the same number and size of blocks");
    kptrs[l] = k.data_ptr();
    vptrs[l] = v.data_ptr();
  }
  std::vector<int64_t> pairs;
  for (const auto& kv : block_mapping) {
    for (int64_t dst : kv.second) {
      TORCH_CHECK(kv.first >= 0 && kv.first < num_blocks && dst >= 0 && dst < num_blocks, what,
                  ": block number out of range: ", kv.first, " -> ", dst);
      pairs.push_back(kv.first);
      pairs.push_back(dst);
    }
  }
  if (pairs.empty()) return;
  PA_DEVICE_GUARD(key_caches[0]);
  check_status(pa_copy_blocks(kptrs.data(), vptrs.data(), num_layers, pairs.data(), static_cast<int>(pairs.size() / 2),
                              block_bytes, current_stream(key_caches[0])),
               what);
}

struct TokenCacheArgs {
  int num_tokens, num_heads, head_size, block_size, dtype;
  int64_t key_stride, value_stride;
};

TokenCacheArgs check_token_cache(const char* what, torch::Tensor& key, torch::Tensor& value, torch::Tensor& key_cache,
                                 torch::Tensor& value_cache, torch::Tensor& slot_mapping) {
  TORCH_CHECK(key.is_cuda() && value.is_cuda() && key_cache.is_cuda() && value_cache.is_cuda() && slot_mapping.is_cuda(),
              what, ": all tensors must be on the GPU");
  TokenCacheArgs a;
  a.dtype = dtype_code(key, what);
  TORCH_CHECK(value.scalar_type() == key.scalar_type() && key_cache.scalar_type() == key.scalar_type() &&
                  value_cache.scalar_type() == key.scalar_type(),
              what, ": key, value and the caches must share one data type");
  check_token_rows(key, "key");
  check_token_rows(value, "value");
  TORCH_CHECK(key.sizes() == value.sizes(), what, ": key and value shapes differ");
  TORCH_CHECK(key_cache.dim() == 5 && value_cache.dim() == 4 && key_cache.is_contiguous() && value_cache.is_contiguous(),
              what, ": key_cache must be a contiguous 5-d tensor and value_cache a contiguous 4-d tensor");
  TORCH_CHECK(slot_mapping.scalar_type() == at::ScalarType::Int && slot_mapping.is_contiguous(), what,
              ": slot_mapping must be a contiguous int32 tensor");
  a.num_tokens = key.size(0);
  a.num_heads = key.size(1);
  a.head_size = key.size(2);
  a.block_size = key_cache.size(3);
  TORCH_CHECK(slot_mapping.numel() >= a.num_tokens, what, ": slot_mapping is shorter than the number of tokens");
  const int64_t x = key_cache.size(4);
  TORCH_CHECK(x * key.element_size() == 16, what, ": key_cache innermost dimension must be 16 bytes");
  TORCH_CHECK(key_cache.size(1) == a.num_heads && key_cache.size(2) * x == a.head_size &&
                  value_cache.size(1) == a.num_heads && value_cache.size(2) == a.head_size &&
                  value_cache.size(3) == a.block_size && value_cache.size(0) == key_cache.size(0),
              what, ": cache shapes do not match key/value");
  a.key_stride = a.num_tokens > 0 ? key.stride(0) : 0;
  a.value_stride = a.num_tokens > 0 ? value.stride(0) : 0;
  return a;
}

// csrc/cache_kernels.cu:186-224
void reshape_and_cache(torch::Tensor& key, torch::Tensor& value, torch::Tensor& key_cache, torch::Tensor& value_cache,
                       torch::Tensor& slot_mapping) {
  const char* what = "reshape_and_cache";
  const TokenCacheArgs a = check_token_cache(what, key, value, key_cache, value_cache, slot_mapping);
  PA_DEVICE_GUARD(key);
  check_status(pa_reshape_and_cache(key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
                                    slot_mapping.data_ptr<int>(), a.num_tokens, a.num_heads, a.head_size, a.block_size,
                                    a.key_stride, a.value_stride, a.dtype, current_stream(key)),
               what);
}

// No counterpart as ONE function in the reference: the pair of calls a rotary attention layer
// makes for the tokens it caches, pos_encoding_ops.rotary_embedding_neox(positions, query, key,
// head_size, cos_sin_cache) then cache_ops.reshape_and_cache(key, value, key_cache, value_cache,
// slot_mapping) (vllm/model_executor/layers/attention.py:200-206, :134-140), with the same
// results in query, key and the caches.  query / key / value are [num_tokens, num_heads, head_size]
// views of the fused projection.
void rotary_embedding_neox_reshape_and_cache(torch::Tensor& positions, torch::Tensor& query, torch::Tensor& key,
                                             torch::Tensor& value, torch::Tensor& key_cache,
                                             torch::Tensor& value_cache, torch::Tensor& slot_mapping,
                                             torch::Tensor& cos_sin_cache) {
  const char* what = "rotary_embedding_neox_reshape_and_cache";
  const TokenCacheArgs a = check_token_cache(what, key, value, key_cache, value_cache, slot_mapping);
  TORCH_CHECK(positions.is_cuda() && query.is_cuda() && cos_sin_cache.is_cuda(), what, ": all tensors must be on the GPU");
  check_token_rows(query, "query");
  TORCH_CHECK(query.sizes() == key.sizes() && query.scalar_type() == key.scalar_type() &&
                  cos_sin_cache.scalar_type() == key.scalar_type(),
              what, ": query, key and cos_sin_cache must share shape / data type");
  TORCH_CHECK(positions.scalar_type() == at::ScalarType::Long && positions.is_contiguous() &&
                  positions.numel() >= a.num_tokens,
              what, ": positions must be a contiguous int64 tensor with one entry per token");
  TORCH_CHECK(cos_sin_cache.dim() == 2 && cos_sin_cache.is_contiguous(), what,
              ": cos_sin_cache must be a contiguous [max_position, rot_dim] tensor");
  PA_DEVICE_GUARD(key);
  check_status(pa_rotary_embedding_neox_reshape_and_cache(
                   positions.data_ptr<int64_t>(), query.data_ptr(), key.data_ptr(), value.data_ptr(),
                   cos_sin_cache.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), slot_mapping.data_ptr<int>(),
                   a.num_tokens, a.num_heads, a.head_size, static_cast<int>(cos_sin_cache.size(1)), a.block_size,
                   a.num_tokens > 0 ? query.stride(0) : 0, a.key_stride, a.value_stride, a.dtype, current_stream(key)),
               what);
}

// csrc/cache_kernels.cu:348-386
void gather_cached_kv(torch::Tensor& key, torch::Tensor& value, torch::Tensor& key_cache, torch::Tensor& value_cache,
                      torch::Tensor& slot_mapping) {
  const char* what = "gather_cached_kv";
  const TokenCacheArgs a = check_token_cache(what, key, value, key_cache, value_cache, slot_mapping);
  PA_DEVICE_GUARD(key);
  check_status(pa_gather_cached_kv(key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
                                   slot_mapping.data_ptr<int>(), a.num_tokens, a.num_heads, a.head_size, a.block_size,
                                   a.key_stride, a.value_stride, a.dtype, current_stream(key)),
               what);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("swap_blocks", &swap_blocks, "Copy whole cache blocks between two cache tensors (GPU<->GPU, GPU<->pinned CPU)");
  m.def("copy_blocks", &copy_blocks, "Copy cache blocks src -> dst inside every layer's key and value cache");
  m.def("reshape_and_cache", &reshape_and_cache, "Scatter new key/value vectors into the paged caches");
  m.def("gather_cached_kv", &gather_cached_kv, "Gather key/value vectors from the paged caches into token rows");
  m.def("rotary_embedding_neox_reshape_and_cache", &rotary_embedding_neox_reshape_and_cache,
        "rotary_embedding_neox on query and key in place, then reshape_and_cache of key and value: one pass");
}
