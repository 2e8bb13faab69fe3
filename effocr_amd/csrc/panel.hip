// Row-panel NT GEMM for the K = embed-dim linears of the ViT encoder (qkv, proj, fc1) on gfx950:
//     out[m][n] = epilogue( sum_k A[m][k] * W[n][k] + bias[n] ),  A = x  or  A = LayerNorm(x)
//
// Why a second GEMM structure: with K = 384 a 128x128 tile has only 6 K-stages, and the K-streaming
// kernel (gemm.hip) pays one full memory latency per stage plus a cold prologue and an epilogue per
// tile (measured: 13.6 us per workgroup for 2.6 us of MFMA work).  Here a workgroup owns 128 token
// rows for the WHOLE output width:
//   * the A panel [128 x K] (bf16/f16, 96 KB at K=384) is loaded ONCE, in a single burst, and stays
//     in LDS; because the workgroup sees complete rows, the pre-norm LayerNorm of timm's Block
//     (norm1 before attn.qkv, norm2 before mlp.fc1) is fused into that load: fp32 residual stream in,
//     fp32 statistics, normalised 16-bit operands straight into LDS — the LayerNorm kernel, its
//     output buffer and one launch per linear disappear;
//   * W (L2-resident, <= 1.2 MB) streams through a 3-deep LDS ring of [256 n x 32 k] stages filled by
//     global_load_lds (16 B/lane, no VGPR round trip) issued TWO stages ahead; waits are counted
//     (s_waitcnt vmcnt(2)) and the workgroup barrier is the raw s_barrier so that the DMA stays in
//     flight across it.  The ring image is lane-linear, so the bank-conflict swizzle
//     (16-B chunk ^= (row>>2)&3) is applied to the per-lane SOURCE address and again on the read;
//   * 8 waves = 2 (token halves) x 4 (64-feature slices) sweep the output in 256-column steps; the
//     MFMA is issued swapped (A-operand = W rows) like gemm.hip, so the epilogue code is shared.
// LDS: 128 x (2K+16) + 3 x 16 KB = 146 KB at K = 384  ->  one workgroup (2 waves/SIMD) per CU.
#include "common.hpp"
#include "kernels.hpp"
#include "ln.hpp"

namespace effocr {
namespace {

constexpr int PBM = 128;            // token rows per panel
constexpr int PNT = 256;            // output columns per sweep step
constexpr int WSTAGE = PNT * 64;    // bytes per ring stage: 256 rows x 32 elements x 2 B
constexpr int RING = 3;

template <typename TO> __device__ __forceinline__ void pstore4(TO* p, float a, float b, float c, float d) {
  if constexpr (sizeof(TO) == 4) {
    f32x4 v = {a, b, c, d};
    *reinterpret_cast<f32x4*>(p) = v;
  } else {
    *reinterpret_cast<u32x2*>(p) = pack4<TO>(a, b, c, d);
  }
}

template <int KD> struct LnShape;                       // LayerNorm lane layout for a row of KD floats
template <> struct LnShape<384> { static constexpr int G = 32, V = 3; };
template <> struct LnShape<128> { static constexpr int G = 32, V = 1; };

template <typename E, int KD, int PRO, int EPI, typename TO>
__global__ __launch_bounds__(512, 2) void panel_gemm_kernel(PanelArgs a) {
  constexpr int APITCH = KD * 2 + 16;                    // bytes per A-panel row (+16: conflict-free b128 reads)
  constexpr int NKS = KD / 32;                           // ring stages per 256-column sweep step
  constexpr int NMAX = 4 * KD;                           // widest layer on this path: mlp.fc1
  __shared__ __attribute__((aligned(16))) char smem[PBM * APITCH + RING * WSTAGE + NMAX * 4];
  char* sA = smem;
  char* sW = smem + PBM * APITCH;
  float* sBias = reinterpret_cast<float*>(smem + PBM * APITCH + RING * WSTAGE);   // bias via LDS: epilogue
                                                         // reads must not share the VM counter with the W DMA
  typedef typename Op16<E>::V8 V8;

  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int wv = wave_id(), wn = wv >> 1, wm = wv & 1;
  const int m0 = blockIdx.x * PBM;
  const int niter = (a.N + PNT - 1) / PNT;
  const int S = niter * NKS;
  const char* Wb = static_cast<const char*>(a.W);

  // ---- W ring fill: stage s = (sweep step it, k-stage ks); 2 x 1 KB DMA pieces per wave per stage
  auto issue_w = [&](int s) {
    if (s >= S) return;
    const int it = s / NKS, ks = s - it * NKS;
    char* dst = sW + (s % RING) * WSTAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p0 = (wv * 2 + i) * 64;
      const int p = p0 + lane;
      const int row = p >> 2;
      const int ch = (p & 3) ^ ((row >> 2) & 3);
      int n = it * PNT + row;
      n = n < a.N ? n : a.N - 1;
      const char* g = Wb + ((size_t)n * KD + ks * 32) * 2 + ch * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(dst + p0 * 16), 16, 0, 0);
    }
  };
  issue_w(0);
  issue_w(1);
  for (int n = tid; n < a.N; n += 512) sBias[n] = a.bias[n];

  // ---- A panel
  if constexpr (PRO == PRO_LN) {
    constexpr int G = LnShape<KD>::G, V = LnShape<KD>::V, RPW = 64 / G;
    const float* X = static_cast<const float*>(a.A);
    const int sub = lane % G;
    constexpr int NP = 16 / RPW;                         // passes: this wave's 16 panel rows
    f32x4 xv[NP][V];
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {              // every load of the wave's 16 rows in flight at once
      int m = m0 + wv * 16 + pass * RPW + lane / G;
      m = m < a.M ? m : a.M - 1;
#pragma unroll
      for (int i = 0; i < V; ++i) xv[pass][i] = *reinterpret_cast<const f32x4*>(X + (int64_t)m * KD + (sub + G * i) * 4);
    }
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      const int row = wv * 16 + pass * RPW + lane / G;
      f32x4 y[V];
      ln_apply<G, V>(xv[pass], sub, a.gamma, a.beta, a.eps, y);
#pragma unroll
      for (int i = 0; i < V; ++i)
        *reinterpret_cast<u32x2*>(sA + row * APITCH + (sub + G * i) * 8) = pack4<E>(y[i][0], y[i][1], y[i][2], y[i][3]);
    }
  } else {
    const E* X = static_cast<const E*>(a.A);
    constexpr int CPR = KD / 8;                          // 16-B chunks per row
    constexpr int NCH = PBM * CPR / 512;
    u32x4 v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + 512 * i;
      const int row = id / CPR, c = id - row * CPR;
      int m = m0 + row;
      m = m < a.M ? m : a.M - 1;
      v[i] = *reinterpret_cast<const u32x4*>(X + (int64_t)m * a.lda + c * 8);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + 512 * i;
      const int row = id / CPR, c = id - row * CPR;
      *reinterpret_cast<u32x4*>(sA + row * APITCH + c * 16) = v[i];
    }
  }
  __syncthreads();                                       // panel visible; also drains stages 0,1 (prologue only)

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int sw = (r31 >> 2) & 3;
  const char* pa = sA + (wm * 64 + r31) * APITCH + half * 16;
  TO* out = static_cast<TO*>(a.out);
  int it = 0, ks = 0;
  for (int s = 0; s < S; ++s) {
    // stage s has landed for this wave's own DMA pieces (one newer stage may stay in flight) ...
    if (s + 1 < S) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... and, past the barrier, for everybody's; all reads of ring slot (s+2)%3 (stage s-1) are done too
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_w(s + 2);

    const int n0 = it * PNT;
    const bool active = (n0 + wn * 64) < a.N;            // wave-uniform: partial last sweep step
    if (active) {
      const char* pw = sW + (s % RING) * WSTAGE + (wn * 64 + r31) * 64;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        V8 wf[2], xf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          wf[i] = *reinterpret_cast<const V8*>(pw + i * 32 * 64 + (((2 * s2 + half) ^ sw) * 16));
          xf[i] = *reinterpret_cast<const V8*>(pa + i * 32 * APITCH + (ks * 2 + s2) * 32);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = Op16<E>::mfma(wf[i], xf[j], acc[i][j]);
      }
    }

    if (ks == NKS - 1) {
      if (active) {
        // phase 1: every load (bias from LDS, residual rows from global) before the first store, so
        // that possibly-aliasing stores (in-place residual) cannot serialise them
        f32x4 bv[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            bv[i][q] = *reinterpret_cast<const f32x4*>(sBias + n0 + wn * 64 + i * 32 + 8 * q + 4 * half);
        int mrow[2];
        bool mok[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int m = m0 + wm * 64 + j * 32 + r31;
          mok[j] = m < a.M;
          mrow[j] = mok[j] ? m : a.M - 1;
        }
        if constexpr (EPI == EPI_BIAS_RESID) {
          f32x4 rv[2][2][4];
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int q = 0; q < 4; ++q)
                rv[i][j][q] = *reinterpret_cast<const f32x4*>(a.resid + (int64_t)mrow[j] * a.ldr + n0 + wn * 64 + i * 32 + 8 * q + 4 * half);
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += rv[i][j][q][e];
        }
        // phase 2: bias (+GELU) and stores
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (mok[j]) {
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
                  v0 = gelu_erf_fast(v0); v1 = gelu_erf_fast(v1); v2 = gelu_erf_fast(v2); v3 = gelu_erf_fast(v3);
                }
                pstore4<TO>(out + (int64_t)mrow[j] * a.ldo + n, v0, v1, v2, v3);
              }
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      ks = 0; ++it;
    } else {
      ++ks;
    }
  }
}

template <typename E, int KD>
int launch_panel(int pro, int epi, const PanelArgs& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.M + PBM - 1) / PBM)), blk(512);
#define EFFOCR_PANEL(P, EP, TOUT) hipLaunchKernelGGL((panel_gemm_kernel<E, KD, P, EP, TOUT>), grid, blk, 0, s, a)
  if (pro == PRO_LN) {
    switch (epi) {
      case EPI_BIAS: EFFOCR_PANEL(PRO_LN, EPI_BIAS, E); break;
      case EPI_BIAS_GELU: EFFOCR_PANEL(PRO_LN, EPI_BIAS_GELU, E); break;
      case EPI_BIAS_RESID: EFFOCR_PANEL(PRO_LN, EPI_BIAS_RESID, float); break;
      default: return fail(EFFOCR_EINVAL, "panel_gemm: unknown epilogue");
    }
  } else {
    switch (epi) {
      case EPI_BIAS: EFFOCR_PANEL(PRO_COPY, EPI_BIAS, E); break;
      case EPI_BIAS_GELU: EFFOCR_PANEL(PRO_COPY, EPI_BIAS_GELU, E); break;
      case EPI_BIAS_RESID: EFFOCR_PANEL(PRO_COPY, EPI_BIAS_RESID, float); break;
      default: return fail(EFFOCR_EINVAL, "panel_gemm: unknown epilogue");
    }
  }
#undef EFFOCR_PANEL
  return check_launch("panel_gemm");
}

}  // namespace

bool panel_gemm_supported(int prec, int N, int K) {
  return (prec == PREC_BF16 || prec == PREC_FP16) && (K == 384 || K == 128) && N > 0 && N % 128 == 0 && N <= 4 * K;
}

int panel_gemm(int prec, int pro, int epi, const PanelArgs& a, hipStream_t s) {
  if (a.M <= 0) return EFFOCR_OK;
  if (!panel_gemm_supported(prec, a.N, a.K)) return fail(EFFOCR_EUNSUPPORTED, "panel_gemm: needs bf16/fp16, K in {128,384}, N % 128 == 0, N <= 4K");
  if (pro == PRO_COPY && (a.lda % 8) != 0) return fail(EFFOCR_EINVAL, "panel_gemm: A rows must be 16-byte aligned");
  if (prec == PREC_BF16) return a.K == 384 ? launch_panel<__bf16, 384>(pro, epi, a, s) : launch_panel<__bf16, 128>(pro, epi, a, s);
  return a.K == 384 ? launch_panel<_Float16, 384>(pro, epi, a, s) : launch_panel<_Float16, 128>(pro, epi, a, s);
}

}  // namespace effocr
