// Glue shared by the pybind11 operator modules: tensor checks, dtype
// mapping, current-stream lookup and error translation.  The modules keep the
// reference's Python-visible names and signatures (csrc/attention.cpp:3-19,
// csrc/cache.cpp:6-47); all device work happens behind the C ABI in
// paged_attn_mi355x.h.
#pragma once

#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include "paged_attn_mi355x.h"

namespace pa_shim {

// The reference launches on at::cuda::getCurrentCUDAStream()
// (attention_kernels.cu:369, cache_kernels.cu:33,127,204,366); CacheEngine
// relies on that to run swaps under torch.cuda.stream(cache_stream).
inline void* current_stream(const torch::Tensor& t) {
  return static_cast<void*>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

// Makes the tensor's device the current one for the duration of a call: the library's
// per-device set-up (kernel attributes, its own scratch) and its launches go by the current
// device, while the stream is looked up from the tensor's device.  The reference's launchers
// have no such guard and misbehave when a tensor lives on a GPU that is not current.
#define PA_DEVICE_GUARD(tensor) const c10::DeviceGuard pa_device_guard_((tensor).device())

inline int dtype_code(const torch::Tensor& t, const char* what) {
  switch (t.scalar_type()) {
    case at::ScalarType::Float: return PA_DTYPE_F32;
    case at::ScalarType::Half: return PA_DTYPE_F16;
    case at::ScalarType::BFloat16: return PA_DTYPE_BF16;
    default: TORCH_CHECK(false, what, ": unsupported data type: ", t.dtype());
  }
  return -1;
}

inline void check_status(int rc, const char* what) {
  TORCH_CHECK(rc == PA_OK, what, " failed (code ", rc, "): ", pa_last_error());
}

// [T, H, D] tensor whose two inner dimensions are dense; the row stride is free.
inline void check_token_rows(const torch::Tensor& t, const char* name) {
  TORCH_CHECK(t.dim() == 3, name, " must be [num_tokens, num_heads, head_size]");
  TORCH_CHECK(t.numel() == 0 || ((t.size(2) == 1 || t.stride(2) == 1) && (t.size(1) == 1 || t.stride(1) == t.size(2))),
              name, " must be dense in its last two dimensions");
}

}  // namespace pa_shim
