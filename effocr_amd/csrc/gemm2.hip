// K-streaming NT GEMM, second generation ("gemm2"): out = epilogue(X[M,K] . W[N,K]^T + bias) for long K
// (mlp.fc2, K = 4*embed) on gfx950, bf16 / f16 operands.
//
// Why: gemm.hip keeps only one register-staged K-stage ahead (32 KB in flight per CU with two
// workgroups) and measured 1.7 us per stage for 0.24 us of MFMA work on fc2 — the X operand (the 620 MB
// MLP hidden tensor) streams from HBM, so the loop runs at memory LATENCY, not bandwidth (Little: 23
// GB/s/CU x ~2 us = 46 KB must be in flight).  Here both operands go through a 3-slot LDS ring filled by
// global_load_lds (no VGPR round trip), two stages = 96 KB in flight per CU, counted vmcnt + raw
// s_barrier so the DMA spans the barrier:
//   * tile 256 tokens x 128 features, one workgroup (8 waves: 4 token quarters x 2 feature halves) per
//     CU; stage = X [256 x 64 k] 32 KB + W [128 x 64 k] 16 KB; 16 MFMAs per wave between barriers;
//   * lane-linear ring image, 128-byte rows, bank swizzle chunk ^= (row>>1)&7 applied on the DMA
//     source address and on the fragment read;
//   * MFMA issued swapped (A-operand = W rows): a lane owns 4 consecutive features of one token;
//     64x64 wave tile = 1 fragment read per MFMA;
//   * the three feature tiles of a token tile are co-scheduled on one XCD (xcd_remap) so X is read
//     from HBM once and twice from that XCD's L2.
#include "common.hpp"
#include "kernels.hpp"


namespace effocr {
namespace {

constexpr int G2M = 256, G2N = 128;
constexpr int G2A = G2M * 128, G2W = G2N * 128, G2STAGE = G2A + G2W;     // bytes
constexpr int G2RING = 3;
constexpr int G2PIECES = G2STAGE / 1024 / 8;                              // 6 DMA pieces per wave per stage

template <typename E, int EPI, typename TO>
__global__ __launch_bounds__(512, 2) void gemm2_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[G2RING * G2STAGE];
  typedef typename Op16<E>::V8 V8;
  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int wv = wave_id(), wn = wv & 1, wm = wv >> 1;
  const int ntn = g.N / G2N;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / ntn) * G2M, n0 = (bid % ntn) * G2N;
  const int nst = g.K / 64;
  const char* Xb = static_cast<const char*>(g.X);
  const char* Wb = static_cast<const char*>(g.W);

  // per-lane DMA sources: piece p (1 KB = 8 rows x 128 B) of the stage image; pieces 0..31 = X, 32..47 = W
  const char* src[G2PIECES];
  int sinc[G2PIECES];                                    // per-stage source advance (blocked X: 8 chunk cells)
#pragma unroll
  for (int i = 0; i < G2PIECES; ++i) {
    const int piece = wv * G2PIECES + i;
    const int p = piece * 64 + lane;
    int row = p >> 3;
    const int ch = (p & 7) ^ ((row >> 1) & 7);
    if (piece < 32) {
      int m = m0 + row;
      m = m < g.M ? m : g.M - 1;
      if (g.blk_x) { src[i] = Xb + blk_off(m, ch, (int)(g.ldx / 8)); sinc[i] = 8 * 512; }
      else { src[i] = Xb + ((size_t)m * g.ldx) * sizeof(E) + ch * 16; sinc[i] = 128; }
    } else {
      row -= 256;
      src[i] = Wb + ((size_t)(n0 + row) * g.ldw) * sizeof(E) + ch * 16;
      sinc[i] = 128;
    }
  }
  auto issue = [&](int s) {
    if (s >= nst) return;
    char* dst = smem + (s % G2RING) * G2STAGE;
#pragma unroll
    for (int i = 0; i < G2PIECES; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)s * sinc[i]),
                                       (__attribute__((address_space(3))) void*)(dst + (wv * G2PIECES + i) * 1024), 16, 0, 0);
  };
  issue(0);
  issue(1);

  f32x16 acc[2][2];                                      // [feature tile][token tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int sw = (r31 >> 1) & 7;
  const int xrow = (wm * 64 + r31) * 128, wrow = G2A + (wn * 64 + r31) * 128;
  auto load_f = [&](V8 (&w)[2], V8 (&x)[2], const char* st, int c4) {
    const int co = ((2 * c4 + half) ^ sw) * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      w[i] = *reinterpret_cast<const V8*>(st + wrow + i * 32 * 128 + co);
      x[i] = *reinterpret_cast<const V8*>(st + xrow + i * 32 * 128 + co);
    }
  };

  // epilogue rows, and the epilogue's global operands (bias, residual / pos-embed): fetched right after
  // the LAST stage's barrier, where the DMA queue is empty, so they land under that stage's MFMAs
  bool mok[2];
  int64_t orow[2];
  const float* posrow[2] = {nullptr, nullptr};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int m = m0 + wm * 64 + j * 32 + r31;
    mok[j] = m < g.M;
    m = mok[j] ? m : g.M - 1;
    orow[j] = m;
    if constexpr (EPI == EPI_PATCH) {
      const int img = m / g.P, p = m - img * g.P;
      orow[j] = (int64_t)img * (g.P + 1) + 1 + p;
      posrow[j] = g.pos + (int64_t)(1 + p) * g.N;
    }
  }
  constexpr bool kAdd = (EPI == EPI_BIAS_RESID || EPI == EPI_PATCH);
  f32x4 bv[2][4];
  f32x4 rv[kAdd ? 2 : 1][2][4];
  auto fetch_epilogue = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[i][q] = *reinterpret_cast<const f32x4*>(g.bias + n0 + wn * 64 + i * 32 + 8 * q + 4 * half);
    if constexpr (kAdd) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn * 64 + i * 32 + 8 * q + 4 * half;
            if constexpr (EPI == EPI_BIAS_RESID) {
              const char* rp = g.blk_out ? reinterpret_cast<const char*>(g.resid) + blk_off(orow[j], n >> 2, g.N >> 2)
                                         : reinterpret_cast<const char*>(g.resid + orow[j] * g.ldr + n);
              rv[i][j][q] = *reinterpret_cast<const f32x4*>(rp);
            }
            else rv[i][j][q] = *reinterpret_cast<const f32x4*>(posrow[j] + n);
          }
    }
  };

  for (int s = 0; s < nst; ++s) {
    // stage s has landed for this wave's own pieces (stage s+1 may stay in flight) ...
    if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ... and past the barrier for everybody's; slot (s+2)%3 = (s-1)%3 is no longer being read
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(s + 2);
    if (s + 1 == nst) fetch_epilogue();
    const char* st = smem + (s % G2RING) * G2STAGE;
    V8 wa[2], xa[2], wb[2], xb[2];
    load_f(wa, xa, st, 0);
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      V8 (&cw)[2] = (c4 & 1) ? wb : wa;
      V8 (&cx)[2] = (c4 & 1) ? xb : xa;
      if (c4 < 3) { if (c4 & 1) load_f(wa, xa, st, c4 + 1); else load_f(wb, xb, st, c4 + 1); }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = Op16<E>::mfma(cw[i], cx[j], acc[i][j]);
        }
    }
  }

  // ---- epilogue (every load was issued in fetch_epilogue, before the first store: out may alias resid)
  TO* out = static_cast<TO*>(g.out);
  if constexpr (kAdd) {
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
          v0 = gelu_erf_fast(v0); v1 = gelu_erf_fast(v1); v2 = gelu_erf_fast(v2); v3 = gelu_erf_fast(v3);
        }
        TO* p = out + orow[j] * g.ldo + n;
        if constexpr (sizeof(TO) == 4) {
          if (g.blk_out) p = reinterpret_cast<TO*>(reinterpret_cast<char*>(out) + blk_off(orow[j], n >> 2, g.N >> 2));
          f32x4 o = {v0, v1, v2, v3};
          *reinterpret_cast<f32x4*>(p) = o;
        }
        else *reinterpret_cast<u32x2*>(p) = pack4<TO>(v0, v1, v2, v3);
      }
    }
  }
}

template <typename E>
int launch2(int epi, const GemmArgs& g, hipStream_t s) {
  const int grid = ((g.M + G2M - 1) / G2M) * (g.N / G2N);
  switch (epi) {
    case EPI_BIAS:       hipLaunchKernelGGL((gemm2_kernel<E, EPI_BIAS, E>), dim3(grid), dim3(512), 0, s, g); break;
    case EPI_BIAS_GELU:  hipLaunchKernelGGL((gemm2_kernel<E, EPI_BIAS_GELU, E>), dim3(grid), dim3(512), 0, s, g); break;
    case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm2_kernel<E, EPI_BIAS_RESID, float>), dim3(grid), dim3(512), 0, s, g); break;
    case EPI_PATCH:      hipLaunchKernelGGL((gemm2_kernel<E, EPI_PATCH, float>), dim3(grid), dim3(512), 0, s, g); break;
    default: return fail(EFFOCR_EINVAL, "gemm2: unknown epilogue");
  }
  return check_launch("gemm2");
}

}  // namespace

bool gemm2_supported(int prec, int N, int K) {
  return (prec == PREC_BF16 || prec == PREC_FP16) && N > 0 && N % G2N == 0 && K >= 128 && K % 64 == 0;
}

int gemm2_nt(int prec, int epi, const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0) return EFFOCR_OK;
  if (!gemm2_supported(prec, g.N, g.K)) return fail(EFFOCR_EUNSUPPORTED, "gemm2: needs bf16/fp16, N % 128 == 0, K % 64 == 0, K >= 128");
  if ((g.ldx * 2) % 16 != 0 || (g.ldw * 2) % 16 != 0) return fail(EFFOCR_EINVAL, "gemm2: operand rows must be 16-byte aligned");
  return prec == PREC_BF16 ? launch2<__bf16>(epi, g, s) : launch2<_Float16>(epi, g, s);
}

}  // namespace effocr
