// Token-stationary row-panel GEMM ("panelr") for the K = embed-dim linears of the ViT encoder:
//     out[m][n] = epilogue( sum_k A[m][k] * W[n][k] + bias[n] ),  A = x  or  A = LayerNorm(x)
//
// Successor of panel.hip, designed from its measurements: in panel.hip the MFMA pipe
// (512 cycles per ring stage), the LDS (1.5 fragment reads per MFMA: ~440 cycles) and the vector-memory
// path were all about equally loaded and, with one 152 KB workgroup of lock-stepped waves per CU, did not
// overlap.  Here the TOKEN operand never touches LDS:
//   * a wave owns 32 tokens for the whole output width and keeps their complete K extent as MFMA
//     B-operand fragments in registers (K/16 x 4 VGPRs = 96 at K = 384).  The fused pre-norm LayerNorm
//     writes its result straight into that register layout: lane (t, h) of the wave holds the k-chunks
//     {16c + 8h .. +7} of token t, i.e. exactly half of the row, so the row statistics are a lane-local
//     sum plus ONE cross-half exchange;
//   * LDS holds only the W ring (4 slots of [128 n x 64 k] = 16 KB, filled by global_load_lds two
//     stages ahead, swizzled on the DMA source address) plus bias/gamma/beta: 73 KB -> TWO independent
//     4-wave workgroups per CU, whose prologues / epilogues / barrier waits overlap each other's MFMAs;
//   * per k16 step a wave reads 4 W fragments (one per 32-feature tile) and issues 4 MFMAs against the
//     resident token fragment: 1 LDS read per MFMA, four independent accumulators;
//   * MFMA issued swapped (A-operand = W rows): a lane owns 4 consecutive features of one token.
// Reference role: attn.qkv (LN1 fused), attn.proj (+residual) and mlp.fc1 (LN2 fused, GELU) of timm's Block.
#include "common.hpp"
#include "kernels.hpp"
#include <type_traits>

namespace effocr {
namespace {

constexpr int RBM = 128;            // token rows per workgroup (4 waves x 32)
constexpr int RNT = 128;            // output columns per sweep step
constexpr int RSTAGE = RNT * 128;   // ring stage: 128 W rows x 64 k x 2 B
constexpr int RRING = 4;
constexpr int RG = 4;               // DMA pieces (1 KB) per wave per stage

template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

template <typename E, int KD, int PRO, int EPI, typename TO, bool FULL>
__global__ __launch_bounds__(256, 2) void panelr_kernel(PanelArgs a) {
  constexpr int NK16 = KD / 16;                          // k16 steps over K (token fragments held in registers)
  constexpr int NKS = KD / 64;                           // ring stages per sweep step
  constexpr int NMAX = 4 * KD;
  __shared__ __attribute__((aligned(16))) char smem[RRING * RSTAGE + NMAX * 4 + 2 * KD * 4];
  char* sW = smem;
  float* sBias = reinterpret_cast<float*>(smem + RRING * RSTAGE);
  float* sGam = sBias + NMAX;
  float* sBet = sGam + KD;
  typedef typename Op16<E>::V8 V8;

  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int wv = wave_id();
  const int m0 = blockIdx.x * RBM + wv * 32;             // this wave's first token
  const int niter = a.N / RNT;
  const int S = niter * NKS;
  const char* Wb = static_cast<const char*>(a.W);

  // ---- W ring fill: 16 pieces of 1 KB (8 rows x 128 B) per stage, 4 per wave
  uint32_t wsrc[RG];
#pragma unroll
  for (int i = 0; i < RG; ++i) {
    const int p = (wv * RG + i) * 64 + lane;
    const int row = p >> 3;
    wsrc[i] = (uint32_t)(row * KD * 2 + (((p & 7) ^ ((row >> 1) & 7)) * 16));
  }
  auto issue_w = [&](int s, int slot) {
    if (s >= S) return;
    const int it = s / NKS, ks = s - it * NKS;
    const char* src = Wb + ((size_t)it * RNT * KD + ks * 64) * 2;
    char* dst = sW + slot * RSTAGE;
#pragma unroll
    for (int i = 0; i < RG; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + wsrc[i]),
                                       (__attribute__((address_space(3))) void*)(dst + (wv * RG + i) * 1024), 16, 0, 0);
  };
  issue_w(0, 0);
  issue_w(1, 1);
  issue_w(2, 2);
  for (int n = tid; n < a.N; n += 256) sBias[n] = a.bias[n];
  if constexpr (PRO == PRO_LN) {
    for (int k = tid; k < KD; k += 256) { sGam[k] = a.gamma[k]; sBet[k] = a.beta[k]; }
  }

  // ---- token fragments (B-operand): xf[c] = 8 consecutive k (16c + 8*half ..) of token m0 + r31
  int mtok = m0 + r31;
  const bool mok = FULL || mtok < a.M;
  mtok = mtok < a.M ? mtok : a.M - 1;
  V8 xf[NK16];
  if constexpr (PRO == PRO_LN) {
    // Two light passes instead of holding the half-row (192 fp32 registers) live, which made the
    // allocator spill long-lived values into the main loop: (1) shifted one-pass statistics
    // (d = x - x[0]; var = E[d^2] - E[d]^2, fp32: the shift removes the cancellation unless x[0] is an
    // extreme outlier of its row), 16 loads in flight; (2) reload (L2 hits), normalise, pack.
    const float* xr = static_cast<const float*>(a.A) + (int64_t)mtok * KD + half * 8;
    const float sh = static_cast<const float*>(a.A)[(int64_t)mtok * KD];
    float s1 = 0.f, s2 = 0.f;
    sfor<0, NK16 / 8>([&](auto Gi) {
      constexpr int g = decltype(Gi)::value;
      f32x4 t[8][2];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        t[c][0] = *reinterpret_cast<const f32x4*>(xr + (g * 8 + c) * 16);
        t[c][1] = *reinterpret_cast<const f32x4*>(xr + (g * 8 + c) * 16 + 4);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d0 = t[c][0][e] - sh, d1 = t[c][1][e] - sh;
          s1 += d0 + d1;
          s2 += d0 * d0 + d1 * d1;
        }
      __builtin_amdgcn_sched_barrier(0);
    });
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float md = s1 * (1.0f / KD);
    const float mean = sh + md;
    const float rstd = 1.0f / sqrtf(fmaxf(s2 * (1.0f / KD) - md * md, 0.f) + a.eps);
    __syncthreads();                                     // gamma / beta / bias visible
    sfor<0, NK16 / 4>([&](auto Gi) {
      constexpr int g = decltype(Gi)::value;
      f32x4 t[4][2];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        t[c][0] = *reinterpret_cast<const f32x4*>(xr + (g * 4 + c) * 16);
        t[c][1] = *reinterpret_cast<const f32x4*>(xr + (g * 4 + c) * 16 + 4);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int cc = g * 4 + c;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(sGam + cc * 16 + half * 8);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(sGam + cc * 16 + half * 8 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sBet + cc * 16 + half * 8);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(sBet + cc * 16 + half * 8 + 4);
        V8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = (E)((t[c][0][e] - mean) * rstd * g0[e] + b0[e]);
          v[4 + e] = (E)((t[c][1][e] - mean) * rstd * g1[e] + b1[e]);
        }
        xf[cc] = v;
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  } else {
    const E* xr = static_cast<const E*>(a.A) + (int64_t)mtok * a.lda + half * 8;
#pragma unroll
    for (int c = 0; c < NK16; ++c) xf[c] = *reinterpret_cast<const V8*>(xr + c * 16);
    __syncthreads();                                     // bias visible
  }

  f32x16 acc[4];                                         // [32-feature tile]: 32 features x 32 tokens
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

  const int sw = (r31 >> 1) & 7;
  const int wrow = r31 * 128;
  TO* out = static_cast<TO*>(a.out);
  const uint32_t prow = (uint32_t)(((int64_t)mtok * a.ldo + 4 * half) * (int64_t)sizeof(TO));
  // W fragments (A-operand) of one k16 step: 4 feature tiles
  auto load_w = [&](V8 (&w)[4], int slot, int c4) {
    const char* pw = sW + slot * RSTAGE + wrow + (((2 * c4 + half) ^ sw) * 16);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) w[nt] = *reinterpret_cast<const V8*>(pw + nt * 32 * 128);
  };

  // epilogue of one sweep step, straight from the accumulators: group (nt, q) = 4 consecutive features
  auto epilogue = [&](int n0) {
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {                     // two batches of 8 groups keep the temporaries small
      float v[32];
      if constexpr (EPI == EPI_BIAS_RESID) {
        f32x4 rv[8];
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) {
          const int nt = hb * 2 + (gg >> 2), q = gg & 3;
          rv[gg] = *reinterpret_cast<const f32x4*>(a.resid + (int64_t)mtok * a.ldr + n0 + nt * 32 + 8 * q + 4 * half);
        }
#pragma unroll
        for (int gg = 0; gg < 8; ++gg)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[gg * 4 + e] = rv[gg][e];
      } else {
#pragma unroll
        for (int e = 0; e < 32; ++e) v[e] = 0.f;
      }
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) {
        const int nt = hb * 2 + (gg >> 2), q = gg & 3;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + n0 + nt * 32 + 8 * q + 4 * half);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[gg * 4 + e] += acc[nt][4 * q + e] + bv[e];
      }
      if constexpr (EPI == EPI_BIAS_GELU) gelu_erf_fast_n<32>(v);
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) {
        const int nt = hb * 2 + (gg >> 2), q = gg & 3;
        TO* p = reinterpret_cast<TO*>(reinterpret_cast<char*>(out) + (prow + (uint32_t)((n0 + nt * 32 + 8 * q) * (int)sizeof(TO))));
        if constexpr (sizeof(TO) == 4) {
          f32x4 o = {v[gg * 4], v[gg * 4 + 1], v[gg * 4 + 2], v[gg * 4 + 3]};
          if (FULL || mok) *reinterpret_cast<f32x4*>(p) = o;
        } else {
          const u32x2 o = pack4<TO>(v[gg * 4], v[gg * 4 + 1], v[gg * 4 + 2], v[gg * 4 + 3]);
          if (FULL || mok) *reinterpret_cast<u32x2*>(p) = o;
        }
      }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  };

  // ---- main loop.  Stage s lives in slot s % 4; its DMA was issued at the top of stage s-3.
  // Top of stage s: stage s+1 must have landed (its first W fragments are prefetched during stage s);
  // the in-order VM counter may still hold stage s+2 (4 pieces) and — for the two stages that follow a
  // sweep-step boundary — the 16 epilogue stores issued after it.
  V8 wa[4], wb[4];
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // stage 0 landed (stages 1, 2 may be in flight)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  load_w(wa, 0, 0);
  int s = 0;
  for (int it = 0; it < niter; ++it) {
    sfor<0, NKS>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      const int slot = s & 3, nslot = (s + 1) & 3;
      if (s + 2 >= S) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (FULL && ks < 2 && it > 0) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // stage s+1 visible; everyone is done with slot (s+3)%4 = (s-1)%4
      asm volatile("" ::: "memory");
      issue_w(s + 3, (s + 3) & 3);
      sfor<0, 4>([&](auto C4) {
        constexpr int c4 = decltype(C4)::value;
        V8 (&cur)[4] = (c4 & 1) ? wb : wa;
        V8 (&nxt)[4] = (c4 & 1) ? wa : wb;
        if constexpr (c4 < 3) load_w(nxt, slot, c4 + 1);
        else load_w(nxt, nslot, 0);                      // first step of stage s+1 (garbage after the last stage)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = Op16<E>::mfma(cur[nt], xf[ks * 4 + c4], acc[nt]);
      });
      ++s;
    });
    epilogue(it * RNT);
  }
}

template <typename E, int KD, bool FULL>
int launch_r(int pro, int epi, const PanelArgs& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.M + RBM - 1) / RBM)), blk(256);
#define EFFOCR_R(P, EP, TOUT) hipLaunchKernelGGL((panelr_kernel<E, KD, P, EP, TOUT, FULL>), grid, blk, 0, s, a)
  if (pro == PRO_LN) {
    switch (epi) {
      case EPI_BIAS: EFFOCR_R(PRO_LN, EPI_BIAS, E); break;
      case EPI_BIAS_GELU: EFFOCR_R(PRO_LN, EPI_BIAS_GELU, E); break;
      case EPI_BIAS_RESID: EFFOCR_R(PRO_LN, EPI_BIAS_RESID, float); break;
      default: return fail(EFFOCR_EINVAL, "panelr_gemm: unknown epilogue");
    }
  } else {
    switch (epi) {
      case EPI_BIAS: EFFOCR_R(PRO_COPY, EPI_BIAS, E); break;
      case EPI_BIAS_GELU: EFFOCR_R(PRO_COPY, EPI_BIAS_GELU, E); break;
      case EPI_BIAS_RESID: EFFOCR_R(PRO_COPY, EPI_BIAS_RESID, float); break;
      default: return fail(EFFOCR_EINVAL, "panelr_gemm: unknown epilogue");
    }
  }
#undef EFFOCR_R
  return check_launch("panelr_gemm");
}

template <typename E, int KD>
int launch_r_full(int pro, int epi, const PanelArgs& a, hipStream_t s) {
  const bool full = (a.M % RBM == 0) || a.rows_padded;
  return full ? launch_r<E, KD, true>(pro, epi, a, s) : launch_r<E, KD, false>(pro, epi, a, s);
}

}  // namespace

int panelr_gemm(int prec, int pro, int epi, const PanelArgs& a, hipStream_t s) {
  if (a.M <= 0) return EFFOCR_OK;
  if (!panel_gemm_supported(prec, a.N, a.K)) return fail(EFFOCR_EUNSUPPORTED, "panelr_gemm: needs bf16/fp16, K in {128,384}, N % 128 == 0, N <= 4K");
  if (pro == PRO_COPY && (a.lda % 8) != 0) return fail(EFFOCR_EINVAL, "panelr_gemm: A rows must be 16-byte aligned");
  if ((int64_t)(a.M + RBM) * a.ldo * 4 >= ((int64_t)1 << 32)) return fail(EFFOCR_EUNSUPPORTED, "panelr_gemm: output larger than 4 GB");
  if (prec == PREC_BF16) return a.K == 384 ? launch_r_full<__bf16, 384>(pro, epi, a, s) : launch_r_full<__bf16, 128>(pro, epi, a, s);
  return a.K == 384 ? launch_r_full<_Float16, 384>(pro, epi, a, s) : launch_r_full<_Float16, 128>(pro, epi, a, s);
}

}  // namespace effocr
