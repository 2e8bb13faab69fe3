// fused MLP (mlp_kernel.hpp) instantiated for __bf16, projection phase true
#include "mlp_kernel.hpp"
namespace effocr {
int mlp_launch_bf16p(const MlpArgs& a, hipStream_t s) { return launch_mlp<__bf16, true>(a, s); }
}  // namespace effocr
#ifdef MLP_STAMP
extern "C" int effocr_debug_mlp_stamps(unsigned long long* out, int n) {
  hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(effocr::mlp_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif
