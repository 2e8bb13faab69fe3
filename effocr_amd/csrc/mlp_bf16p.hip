// fused MLP (mlp_kernel.hpp) instantiated for __bf16, projection phase true
#include "mlp_kernel.hpp"
namespace effocr {
int mlp_launch_bf16p(const MlpArgs& a, hipStream_t s) { return launch_mlp<__bf16, true>(a, s); }
}  // namespace effocr
