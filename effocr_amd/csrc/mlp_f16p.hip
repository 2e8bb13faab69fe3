// fused MLP (mlp_kernel.hpp) instantiated for _Float16, projection phase true
#include "mlp_kernel.hpp"
namespace effocr {
int mlp_launch_f16p(const MlpArgs& a, hipStream_t s) { return launch_mlp<_Float16, true>(a, s); }
}  // namespace effocr
