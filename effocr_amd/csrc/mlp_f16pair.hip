// fused MLP (mlp_kernel.hpp): the 64-token wave-pair kernels (whole panels + 6- / 3- / 2-way pair parts) instantiated for _Float16
#include "mlp_kernel.hpp"
namespace effocr {
int mlp_pair_launch_f16(const MlpArgs& a, int tncw, unsigned grid, hipStream_t s) { return launch_mlp_pair<_Float16>(a, tncw, grid, s); }
}  // namespace effocr
