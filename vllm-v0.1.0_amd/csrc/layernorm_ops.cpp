// Python module `layernorm_ops`: same name, function and argument order as
// vLLM v0.1.0's csrc/layernorm.cpp:3-14 (used by
// vllm/model_executor/layers/layernorm.py).
#include "torch_shim.h"

namespace {

void rms_norm(torch::Tensor& out,     // [num_tokens, hidden_size]
              torch::Tensor& input,   // [num_tokens, hidden_size]
              torch::Tensor& weight,  // [hidden_size]
              float epsilon) {
  using namespace pa_shim;
  const char* what = "rms_norm";
  TORCH_CHECK(out.is_cuda() && input.is_cuda() && weight.is_cuda(), what, ": all tensors must be on the GPU");
  const int dtype = dtype_code(input, what);
  TORCH_CHECK(out.scalar_type() == input.scalar_type() && weight.scalar_type() == input.scalar_type(), what,
              ": out, input and weight must share one data type");
  TORCH_CHECK(input.dim() == 2 && input.is_contiguous() && out.is_contiguous() && out.sizes() == input.sizes(), what,
              ": out and input must be contiguous [num_tokens, hidden_size] tensors of one shape");
  TORCH_CHECK(weight.is_contiguous() && weight.numel() == input.size(1), what, ": weight must be [hidden_size]");
  PA_DEVICE_GUARD(input);
  check_status(pa_rms_norm(out.data_ptr(), input.data_ptr(), weight.data_ptr(), epsilon,
                           static_cast<int>(input.size(0)), static_cast<int>(input.size(1)), dtype,
                           current_stream(input)),
               what);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("rms_norm", &rms_norm, "Apply Root Mean Square (RMS) Normalization to the input tensor");
}
