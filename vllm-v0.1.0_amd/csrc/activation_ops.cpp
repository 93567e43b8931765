// Python module `activation_ops`: same name, function and argument order as
// vLLM v0.1.0's csrc/activation.cpp:3-12 (used by
// vllm/model_executor/layers/activation.py).
#include "torch_shim.h"

namespace {

void silu_and_mul(torch::Tensor& out,    // [num_tokens, d]
                  torch::Tensor& input)  // [num_tokens, 2 * d]
{
  using namespace pa_shim;
  const char* what = "silu_and_mul";
  TORCH_CHECK(out.is_cuda() && input.is_cuda(), what, ": all tensors must be on the GPU");
  const int dtype = dtype_code(input, what);
  TORCH_CHECK(out.scalar_type() == input.scalar_type(), what, ": out and input must share one data type");
  TORCH_CHECK(input.dim() == 2 && out.dim() == 2 && input.is_contiguous() && out.is_contiguous(), what,
              ": out and input must be contiguous 2-d tensors");
  TORCH_CHECK(input.size(1) % 2 == 0 && out.size(0) == input.size(0) && out.size(1) * 2 == input.size(1), what,
              ": input must be [num_tokens, 2 * d] and out [num_tokens, d]");
  PA_DEVICE_GUARD(input);
  check_status(pa_silu_and_mul(out.data_ptr(), input.data_ptr(), static_cast<int>(input.size(0)),
                               static_cast<int>(out.size(1)), dtype, current_stream(input)),
               what);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("silu_and_mul", &silu_and_mul, "Activation function used in SwiGLU");
}
