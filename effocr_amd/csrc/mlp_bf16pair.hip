// fused MLP (mlp_kernel.hpp): the 64-token wave-pair kernels (whole panels + 6- / 3- / 2-way pair parts) instantiated for __bf16
#include "mlp_kernel.hpp"
namespace effocr {
int mlp_pair_launch_bf16(const MlpArgs& a, int tncw, unsigned grid, hipStream_t s) { return launch_mlp_pair<__bf16>(a, tncw, grid, s); }
}  // namespace effocr
#ifdef MLP_STAMP
// (the stamp table is per translation unit: tools/mlp_timeline.py --pair and tools/mlp_part_timeline.py read the pair kernels' here;
//  build: tools/ab_build.sh stamp "-DMLP_STAMP" mlp_bf16p.hip mlp_bf16pair.hip)
extern "C" int effocr_debug_mlp_pair_stamps(unsigned long long* out, int n) {
  hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(effocr::mlp_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif
