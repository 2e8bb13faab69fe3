// PRODUCT build only (the default `make`): the entry points of the translation units that are reachable ONLY through A/B switches —
// the row-panel GEMM (panel.hip: use_qkvattn = 0, use_projf = 0, use_mlp = 0, use_blocked = 0, effocr_op_linear / effocr_op_ln_linear at
// ViT-S widths) and the fused MLP without the projection phase (mlp_bf16.hip, mlp_f16.hip: use_projf = 0, effocr_op_mlp_blocked,
// effocr_op_mlp_ln_blocked) — fail loudly here.  `make AB=1` links the real kernels into libeffocr_hip_ab.so instead of this file
// (3.0 MB of device code the default dispatch can never reach; tests/ that A/B those paths load that build).
#include "common.hpp"
#include "kernels.hpp"

namespace effocr {

int panel_gemm(int, int, int, const PanelArgs&, hipStream_t) {
  return fail(EFFOCR_EUNSUPPORTED, "row-panel GEMM: an A/B path, not part of the product library (build and load libeffocr_hip_ab.so: make -C effocr_amd/csrc AB=1)");
}
int mlp_launch_bf16(const MlpArgs&, hipStream_t) {
  return fail(EFFOCR_EUNSUPPORTED, "fused MLP without the projection phase: an A/B path, not part of the product library (make -C effocr_amd/csrc AB=1)");
}
int mlp_launch_f16(const MlpArgs&, hipStream_t) {
  return fail(EFFOCR_EUNSUPPORTED, "fused MLP without the projection phase: an A/B path, not part of the product library (make -C effocr_amd/csrc AB=1)");
}

}  // namespace effocr
