// fused MLP (mlp_kernel.hpp) instantiated for _Float16, projection phase false
#include "mlp_kernel.hpp"
namespace effocr {
int mlp_launch_f16(const MlpArgs& a, hipStream_t s) { return launch_mlp<_Float16, false>(a, s); }
}  // namespace effocr
