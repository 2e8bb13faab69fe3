// Fused transformer MLP: dispatcher (kernel and launcher in mlp_kernel.hpp, instantiated in mlp_bf16.hip, mlp_bf16p.hip,
// mlp_f16.hip, mlp_f16p.hip — "p" = with the attention projection phase in front).
#include "common.hpp"
#include "kernels.hpp"

namespace effocr {

int mlp_launch_bf16(const MlpArgs& a, hipStream_t s);
int mlp_launch_bf16p(const MlpArgs& a, hipStream_t s);
int mlp_launch_f16(const MlpArgs& a, hipStream_t s);
int mlp_launch_f16p(const MlpArgs& a, hipStream_t s);

// Width class of the row-panel / per-image / fused-MLP kernels (ViT-S and the miniature test width): 16-bit operands, K in {128, 384}.
// Declared next to panel_gemm (kernels.hpp) because the A/B row-panel GEMM has exactly this domain; the forward uses it to choose
// the ViT-S kernel family.
bool panel_gemm_supported(int prec, int N, int K) {
  return (prec == PREC_BF16 || prec == PREC_FP16) && (K == 384 || K == 128) && N > 0 && N % 128 == 0 && N <= 4 * K;
}

bool mlp_fused_supported(int prec, int D, int H) {
  return (prec == PREC_BF16 || prec == PREC_FP16) && ((D == 384 && H == 1536) || (D == 128 && H == 512));
}

int mlp_fused(int prec, const MlpArgs& a, hipStream_t s) {
  if (a.M <= 0) return EFFOCR_OK;
  if (!mlp_fused_supported(prec, a.D, a.H)) return fail(EFFOCR_EUNSUPPORTED, "mlp_fused: needs bf16/fp16 and (D, H) in {(384, 1536), (128, 512)}");
  if (a.rows_alloc % 32 != 0 || a.rows_alloc < a.M) return fail(EFFOCR_EINVAL, "mlp_fused: rows_alloc must be a multiple of 32 covering M");
  const bool proj = a.A != nullptr;
  if (proj && (!a.Wpp || !a.bp)) return fail(EFFOCR_EINVAL, "mlp_fused: the projection needs its weight and bias");
  if (!a.b2_logical) return fail(EFFOCR_EINVAL, "mlp_fused: the unpermuted fc2 bias is required");
  if (a.xn_out && (!a.gamma_n || !a.beta_n)) return fail(EFFOCR_EINVAL, "mlp_fused: the second output needs the next norm's weight and bias");
  if (prec == PREC_BF16) return proj ? mlp_launch_bf16p(a, s) : mlp_launch_bf16(a, s);
  return proj ? mlp_launch_f16p(a, s) : mlp_launch_f16(a, s);
}

}  // namespace effocr
