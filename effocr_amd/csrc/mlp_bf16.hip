// fused MLP (mlp_kernel.hpp) instantiated for __bf16, projection phase false
#include "mlp_kernel.hpp"
namespace effocr {
int mlp_launch_bf16(const MlpArgs& a, hipStream_t s) { return launch_mlp<__bf16, false>(a, s); }
}  // namespace effocr
