// K-streaming NT GEMM, third generation ("gemm3"): out = epilogue(X[M,K] . W[N,K]^T + bias), bf16 / f16
// operands, BOTH operands and the output in the fragment-blocked layout (common.hpp blk_off).
//
// Why (profiles/README.md, "library yardstick"): gemm2's 64x64 wave tiles need one LDS fragment read and
// 1.5 KB of DMA per MFMA; the vendor library's kernels for these shapes use 128x128 wave tiles (one wave
// per SIMD, 256 accumulators) and run 1.4-1.7x faster.  Same idea here, written for the blocked layout:
//   * tile 256 tokens x TN features (TN = 256, or 192 when N is only a multiple of 192), 4 waves = 2 token
//     halves x 2 feature halves, wave tile 128 x TN/2 = 4 x NT MFMA tiles (NT = TN/64): per k16 step
//     (4 + NT) fragment reads feed 4*NT MFMAs (0.5-0.58 reads/MFMA), 192-256 fp32 accumulators;
//   * stage = 32 k: X 16 KB + W TN*64 B, 4-slot ring filled by global_load_lds three stages ahead
//     (84-96 KB in flight per CU).  A blocked cell [32 rows][16 B] is 512 contiguous bytes in HBM and is
//     copied verbatim: a fragment read (32 rows x 16 B per half-wave) is one contiguous KB -> conflict-free
//     without swizzles, and every DMA lane-group reads 512 contiguous bytes (the fast TA case);
//   * one wave per SIMD means nothing else hides a wave's own latencies, so the barrier sits in the MIDDLE
//     of a stage: [read frags(s, k16=1)] [MFMAs k16=0] [wait stage s+1, barrier, DMA stage s+4 into the
//     slot just vacated, read frags(s+1, k16=0)] [MFMAs k16=1] — both fragment reads fly under MFMAs;
//   * MFMA issued swapped (A-operand = W rows): a lane owns 4 consecutive features of one token.
#include "common.hpp"
#include "kernels.hpp"

namespace effocr {
namespace {

constexpr int G3M = 256, G3RING = 4;

template <typename E, int NT, int EPI, typename TO>
__global__ __launch_bounds__(256, 1) void gemm3_kernel(GemmArgs g) {
  constexpr int TN = NT * 64;
  constexpr int XS = G3M * 64, WS = TN * 64, STAGE = XS + WS;          // bytes per 32-k stage
  constexpr int PX = XS / 1024, PW = WS / 1024, PP = (PX + PW) / 4;    // 1 KB DMA pieces: X, W, per wave
  static_assert((PX + PW) % 4 == 0, "pieces must split evenly over 4 waves");
  __shared__ __attribute__((aligned(16))) char smem[G3RING * STAGE];
  typedef typename Op16<E>::V8 V8;
  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int wv = wave_id(), wm = wv & 1, wn = wv >> 1;
  const int ntn = g.N / TN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / ntn;
  const int m0 = mt * G3M, n0 = (bid - mt * ntn) * TN;
  const int nst = g.K >> 5;
  const int kch = g.K >> 3;                                            // 16-byte chunks per operand row
  const int last_rb = (g.rows_alloc >> 5) - 1;                         // last addressable X row block

  // per-lane DMA sources; piece q (1 KB = two adjacent 16-B chunk cells of one 32-row block) lands at q KB
  const char* src[PP];
#pragma unroll
  for (int i = 0; i < PP; ++i) {
    const int q = wv * PP + i;
    if (q < PX) {
      int rb = (m0 >> 5) + (q >> 1);
      rb = rb < last_rb ? rb : last_rb;                                // rows past the buffer: any valid block
      src[i] = static_cast<const char*>(g.X) + ((size_t)rb * kch + 2 * (q & 1)) * 512 + lane * 16;
    } else {
      const int qq = q - PX;
      src[i] = static_cast<const char*>(g.Wblk) + ((size_t)((n0 >> 5) + (qq >> 1)) * kch + 2 * (qq & 1)) * 512 + lane * 16;
    }
  }
  auto issue = [&](int s) {
    if (s >= nst) return;
    char* dst = smem + (s & (G3RING - 1)) * STAGE + wv * PP * 1024;
#pragma unroll
    for (int i = 0; i < PP; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)s * 2048),
                                       (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
  };
  issue(0); issue(1); issue(2); issue(3);

  f32x16 acc[NT][4];                                                   // [feature tile][token tile]
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int xo = (wm * 16 + half) * 512 + r31 * 16;                    // (row block wm*4 + j, chunk 2*c4 + half)
  const int wo = XS + (wn * NT * 4 + half) * 512 + r31 * 16;
  struct Frags { V8 w[NT]; V8 x[4]; };
  auto load_f = [&](Frags& f, const char* st, int c4) {
#pragma unroll
    for (int i = 0; i < NT; ++i) f.w[i] = *reinterpret_cast<const V8*>(st + wo + (i * 4 + 2 * c4) * 512);
#pragma unroll
    for (int j = 0; j < 4; ++j) f.x[j] = *reinterpret_cast<const V8*>(st + xo + (j * 4 + 2 * c4) * 512);
  };
  auto mma = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = Op16<E>::mfma(f.w[i], f.x[j], acc[i][j]);
  };

  Frags fa, fb;
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PP) : "memory");        // stage 0 (own pieces) ...
  __builtin_amdgcn_s_barrier();                                        // ... and everybody's
  asm volatile("" ::: "memory");
  load_f(fa, smem, 0);

  for (int s = 0; s < nst; ++s) {
    const char* st = smem + (s & (G3RING - 1)) * STAGE;
    load_f(fb, st, 1);
    mma(fa);
    // stage s+1 landed (own pieces; s+2, s+3 may stay in flight), every wave holds its stage-s fragments
    // in registers -> past the barrier slot s&3 is free for stage s+4
    if (s + 3 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(s + 4);
    if (s + 1 < nst) load_f(fa, smem + ((s + 1) & (G3RING - 1)) * STAGE, 0);
    mma(fb);
  }

  // ---- epilogue: lane = token (m0 + wm*128 + j*32 + r31), 4 consecutive features per (i, q)
  TO* out = static_cast<TO*>(g.out);
  constexpr int CH = 16 / (int)sizeof(TO);                             // elements per 16-byte output chunk
  const int nb = n0 + wn * NT * 32 + 4 * half;
  f32x4 bv[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[i][q] = *reinterpret_cast<const f32x4*>(g.bias + nb + i * 32 + 8 * q);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 128 + j * 32 + r31;
    const bool ok = m < g.M;
    const int mr = ok ? m : g.M - 1;
    if constexpr (EPI == EPI_BIAS_RESID) {
      f32x4 rv[NT][4];
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rv[i][q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(g.resid) + blk_off(mr, (nb + i * 32 + 8 * q) >> 2, g.N >> 2));
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += rv[i][q][e];
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = acc[i][j][4 * q + e] + bv[i][q][e];
      if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = (float)(E)v[e];              // same argument rounding as panel.hip
        gelu_erf_fast_n<16>(v);
      }
      if (ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + i * 32 + 8 * q;
          char* p = reinterpret_cast<char*>(out) + blk_off(mr, n / CH, g.N / CH) + (n % CH) * (int)sizeof(TO);
          if constexpr (sizeof(TO) == 4) {
            const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            *reinterpret_cast<f32x4*>(p) = o;
          } else {
            *reinterpret_cast<u32x2*>(p) = pack4<TO>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
          }
        }
      }
    }
  }
}

template <typename E, int NT>
int launch3(int epi, const GemmArgs& g, hipStream_t s) {
  const int grid = ((g.M + G3M - 1) / G3M) * (g.N / (NT * 64));
  switch (epi) {
    case EPI_BIAS:       hipLaunchKernelGGL((gemm3_kernel<E, NT, EPI_BIAS, E>), dim3(grid), dim3(256), 0, s, g); break;
    case EPI_BIAS_GELU:  hipLaunchKernelGGL((gemm3_kernel<E, NT, EPI_BIAS_GELU, E>), dim3(grid), dim3(256), 0, s, g); break;
    case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm3_kernel<E, NT, EPI_BIAS_RESID, float>), dim3(grid), dim3(256), 0, s, g); break;
    default: return fail(EFFOCR_EINVAL, "gemm3: unknown epilogue");
  }
  return check_launch("gemm3");
}

}  // namespace

bool gemm3_supported(int prec, int N, int K) {
  return (prec == PREC_BF16 || prec == PREC_FP16) && N > 0 && (N % 256 == 0 || N % 192 == 0) && K >= 128 && K % 32 == 0;
}

int gemm3_nt(int prec, int epi, const GemmArgs& g_in, hipStream_t s) {
  if (g_in.M <= 0) return EFFOCR_OK;
  if (!gemm3_supported(prec, g_in.N, g_in.K)) return fail(EFFOCR_EUNSUPPORTED, "gemm3: needs bf16/fp16, N % 192 == 0 or N % 256 == 0, K % 32 == 0, K >= 128");
  if (!g_in.Wblk || !g_in.blk_x || !g_in.blk_out) return fail(EFFOCR_EINVAL, "gemm3: operands and output must be fragment-blocked");
  GemmArgs g = g_in;
  if (g.rows_alloc <= 0) g.rows_alloc = ((g.M + 31) / 32) * 32;
  if (g.rows_alloc % 32 != 0 || g.rows_alloc < g.M) return fail(EFFOCR_EINVAL, "gemm3: rows_alloc must be a multiple of 32 covering M");
  const bool wide = g.N % 256 == 0;
  if (prec == PREC_BF16) return wide ? launch3<__bf16, 4>(epi, g, s) : launch3<__bf16, 3>(epi, g, s);
  return wide ? launch3<_Float16, 4>(epi, g, s) : launch3<_Float16, 3>(epi, g, s);
}

}  // namespace effocr
