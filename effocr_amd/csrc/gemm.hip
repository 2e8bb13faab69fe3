// NT GEMM for the recognizer encoder's linear layers on gfx950 MFMA:
//     out[m][n] = epilogue( sum_k X[m][k] * W[n][k] + bias[n] )
// X = activations [M,K] row-major, W = torch Linear weight [N,K] row-major (both K-contiguous, the
// layout the MFMA operand fragments want — no transposes anywhere).
//
// Reference role: every nn.Linear inside timm's VisionTransformer that models/encoders.py:58
// instantiates and infer_effocr.py:314 runs (qkv, proj, fc1(+GELU), fc2, and the 16x16/16
// patch-embedding conv, which is a GEMM over im2col rows).
//
// Structure (v1, "128x128x(128 B)" register-staged double-buffered tile):
//   * 256 threads = 4 waves as 2 (feature) x 2 (token); each wave owns a 64x64 sub-tile as 2x2
//     32x32 MFMA tiles.  The MFMA is issued "swapped": A-operand = W rows (features), B-operand =
//     X rows (tokens), so that in the C/D fragment every lane holds 4 CONSECUTIVE FEATURES of one
//     token -> 8/16-byte row-major epilogue stores and a float4 bias/residual access per group.
//   * one K-stage = 128 bytes of every row (64 bf16/f16 or 32 fp32); LDS rows are padded to 144 B
//     so that the 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots (conflict-free).
//   * global -> registers -> LDS with the loads of stage t+1 in flight under the MFMAs of stage t
//     (one barrier per stage).
//   * operand types: bf16 / f16 (v_mfma_f32_32x32x16_*) and fp32 (v_mfma_f32_32x32x2_f32, exact
//     fp32 fmaf chain, used by the fp32 parity mode).  fp32 stages are stored as two "parity
//     planes" per row (even k | odd k) so that the 32x32x2 MFMA — lanes 0-31 supply k, lanes 32-63
//     supply k+1 — walks k in ascending order with 16-byte LDS reads.
#include "common.hpp"
#include "kernels.hpp"
#include "tile128.hpp"

namespace effocr {

namespace {

using namespace tile128;

template <typename TO> __device__ __forceinline__ void store4(TO* p, float a, float b, float c, float d) {
  if constexpr (sizeof(TO) == 4) {
    f32x4 v = {a, b, c, d};
    *reinterpret_cast<f32x4*>(p) = v;
  } else {
    *reinterpret_cast<u32x2*>(p) = pack4<TO>(a, b, c, d);
  }
}

template <typename TA, int EPI, typename TO>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[GEMM_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = wave_id(), wn = w >> 1, wm = w & 1;
  const int ntn = g.N / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
  const TA* X = static_cast<const TA*>(g.X);
  const TA* W = static_cast<const TA*>(g.W);
  const int nkt = (g.K * (int)sizeof(TA)) / ROWB;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 rw[4], rx[4];
  stage_load<TA>(rw, W, g.ldw, n0, g.N, 0, tid);
  stage_load<TA>(rx, X, g.ldx, m0, g.M, 0, tid);
  stage_store<TA>(rw, smem, tid);
  stage_store<TA>(rx, smem + TILEB, tid);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    char* cur = smem + (kt & 1) * STAGEB;
    char* nxt = smem + ((kt & 1) ^ 1) * STAGEB;
    const bool more = (kt + 1) < nkt;
    if (more) {
      stage_load<TA>(rw, W, g.ldw, n0, g.N, (kt + 1) * ROWB, tid);
      stage_load<TA>(rx, X, g.ldx, m0, g.M, (kt + 1) * ROWB, tid);
    }
    stage_mma<TA>(acc, cur, cur + TILEB, wn, wm, lane);
    if (more) {
      stage_store<TA>(rw, nxt, tid);
      stage_store<TA>(rx, nxt + TILEB, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds, per (i,j,q), 4 consecutive features of one token.  All loads (bias,
  // residual, pos_embed) are issued before the first store: out may alias resid (in-place residual
  // stream), which would otherwise force load -> wait -> store serialisation.
  const int half = lane >> 5;
  TO* out = static_cast<TO*>(g.out);
  f32x4 bv[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[i][q] = *reinterpret_cast<const f32x4*>(g.bias + n0 + wn * 64 + i * 32 + 8 * q + 4 * half);
  bool mok[2];
  int64_t orow[2];
  const float* posrow[2] = {nullptr, nullptr};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int m = m0 + wm * 64 + j * 32 + (lane & 31);
    mok[j] = m < g.M;
    m = mok[j] ? m : g.M - 1;
    orow[j] = m;
    if constexpr (EPI == EPI_PATCH) {
      const int img = m / g.P, p = m - img * g.P;
      orow[j] = (int64_t)img * (g.P + 1) + 1 + p;
      posrow[j] = g.pos + (int64_t)(1 + p) * g.N;
    }
  }
  if constexpr (EPI == EPI_BIAS_RESID || EPI == EPI_PATCH) {
    f32x4 rv[2][2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 64 + i * 32 + 8 * q + 4 * half;
          if constexpr (EPI == EPI_BIAS_RESID) rv[i][j][q] = *reinterpret_cast<const f32x4*>(g.resid + orow[j] * g.ldr + n);
          else rv[i][j][q] = *reinterpret_cast<const f32x4*>(posrow[j] + n);
        }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += rv[i][j][q][e];
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (!mok[j]) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + i * 32 + 8 * q + 4 * half;
        float v0 = acc[i][j][4 * q + 0] + bv[i][q][0];
        float v1 = acc[i][j][4 * q + 1] + bv[i][q][1];
        float v2 = acc[i][j][4 * q + 2] + bv[i][q][2];
        float v3 = acc[i][j][4 * q + 3] + bv[i][q][3];
        if constexpr (EPI == EPI_BIAS_GELU) {
          if constexpr (sizeof(TO) == 4) {
            v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
          } else {
            v0 = gelu_erf_fast(v0); v1 = gelu_erf_fast(v1); v2 = gelu_erf_fast(v2); v3 = gelu_erf_fast(v3);
          }
        }
        store4<TO>(out + orow[j] * g.ldo + n, v0, v1, v2, v3);
      }
    }
  }
}

template <typename TA>
int launch_typed(int epi, const GemmArgs& g, hipStream_t s) {
  const int grid = ((g.M + BM - 1) / BM) * (g.N / BN);
  switch (epi) {
    case EPI_BIAS:       hipLaunchKernelGGL((gemm_nt_kernel<TA, EPI_BIAS, TA>), dim3(grid), dim3(256), 0, s, g); break;
    case EPI_BIAS_GELU:  hipLaunchKernelGGL((gemm_nt_kernel<TA, EPI_BIAS_GELU, TA>), dim3(grid), dim3(256), 0, s, g); break;
    case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm_nt_kernel<TA, EPI_BIAS_RESID, float>), dim3(grid), dim3(256), 0, s, g); break;
    case EPI_PATCH:      hipLaunchKernelGGL((gemm_nt_kernel<TA, EPI_PATCH, float>), dim3(grid), dim3(256), 0, s, g); break;
    default: return fail(EFFOCR_EINVAL, "gemm_nt: unknown epilogue");
  }
  return check_launch("gemm_nt");
}

}  // namespace

int gemm_nt(int prec, int epi, const GemmArgs& g, hipStream_t s) {
  const int es = (prec == PREC_FP32) ? 4 : 2;
  if (g.M <= 0) return EFFOCR_OK;
  if (g.N % BN != 0 || (g.K * es) % ROWB != 0 || g.K <= 0)
    return fail(EFFOCR_EUNSUPPORTED, "gemm_nt: N must be a multiple of 128 and K*elem_size a multiple of 128 bytes");
  if ((g.ldx * es) % 16 != 0 || (g.ldw * es) % 16 != 0)
    return fail(EFFOCR_EINVAL, "gemm_nt: operand rows must be 16-byte aligned");
  switch (prec) {
    case PREC_BF16: return launch_typed<__bf16>(epi, g, s);
    case PREC_FP16: return launch_typed<_Float16>(epi, g, s);
    case PREC_FP32: return launch_typed<float>(epi, g, s);
  }
  return fail(EFFOCR_EINVAL, "gemm_nt: unknown precision");
}

}  // namespace effocr
