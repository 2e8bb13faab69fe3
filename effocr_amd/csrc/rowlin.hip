// Row-block linear layers over the fragment-blocked layout (bf16 / f16 operands, gfx950) — phase A of mlp.hip as a
// kernel of its own, for the two K = embed-dim linears of a transformer block:
//   MODE_LN    out_blk(16-bit) = LayerNorm(x_blk fp32) . W^T + bias          (norm1 + attn.qkv)
//   MODE_RESID x_blk(fp32)    += a_blk(16-bit) . W^T + bias                   (attn.proj + residual)
// One workgroup = 4 waves (one per SIMD, accumulators in AGPRs) = 128 tokens; wave w owns row block w (32 tokens):
// its D/16 B-operand fragments (LayerNorm output, or the 16-bit input rows loaded straight in fragment layout)
// live in registers for the whole sweep over N, so LDS holds nothing but the weight ring and the parameters.
// Weights (fragment-blocked copy) stream through an 8-slot ring of 16 KB stages (4 row blocks x 64 k) filled by
// global_load_lds six stages ahead; barrier in the middle of a stage; one rolling fragment set (see mlp.hip).
// Per chunk of 128 output features: D/64 stages x 16 MFMAs per wave, then
//   MODE_LN    bias + round -> 16 packed registers, stored (8 B per lane, whole 512-B cells per wave) in the MFMA
//              shadows of the NEXT chunk.  Stores share the in-order VM counter with the ring's DMA: the stage
//              waits keep their constant count, which is then merely stricter (a store issued inside the window
//              stands in for a DMA piece), never looser;
//   MODE_RESID the residual rows are requested at the start of the chunk and added / stored at its end.
#include "common.hpp"
#include "kernels.hpp"
#include <type_traits>

namespace effocr {
namespace {

template <int I, int N, typename F> __device__ __forceinline__ void rfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    rfor<I + 1, N>(f);
  }
}

constexpr int RL_STAGE = 16384, RL_RING = 8;

template <typename E, int D, int N, int MODE>
__global__ __launch_bounds__(256, 1) void rowlin_kernel(RowLinArgs a) {
  typedef typename Op16<E>::V8 V8;
  constexpr int KC = D / 8, NC = N / 128, SA = D / 64, NS = NC * SA, NXF = D / 16, R = RL_RING;
  static_assert(D % 64 == 0 && N % 128 == 0, "rowlin: D % 64, N % 128");
  __shared__ __attribute__((aligned(16))) char smem[R * RL_STAGE + (N + 2 * D) * 4];
  char* sW = smem;
  float* sB = reinterpret_cast<float*>(smem + R * RL_STAGE);
  float* sG = sB + N;
  float* sBt = sG + D;

  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int w = wave_id();
  const int64_t rb = (int64_t)blockIdx.x * 4 + w;
  const int64_t rbc = rb;                                // rows_alloc covers whole 128-row panels (checked by the launcher)
  const char* W = static_cast<const char*>(a.Wb);

  // ---- input rows first (oldest in the VM queue)
  V8 xf[NXF];
  f32x4 xv[MODE == ROWLIN_LN ? 2 * NXF : 1];
  if constexpr (MODE == ROWLIN_LN) {
    const char* xb = reinterpret_cast<const char*>(a.x) + rbc * (D / 4) * 512 + r31 * 16;
#pragma unroll
    for (int t = 0; t < NXF; ++t) {
      xv[2 * t] = *reinterpret_cast<const f32x4*>(xb + (size_t)(4 * t + 2 * half) * 512);
      xv[2 * t + 1] = *reinterpret_cast<const f32x4*>(xb + (size_t)(4 * t + 2 * half + 1) * 512);
    }
  } else {
    const char* ab = static_cast<const char*>(a.A) + rbc * KC * 512 + half * 512 + r31 * 16;
#pragma unroll
    for (int t = 0; t < NXF; ++t) xf[t] = *reinterpret_cast<const V8*>(ab + (size_t)t * 1024);
  }
  for (int n = tid; n < N; n += 256) sB[n] = a.bias[n];
  if constexpr (MODE == ROWLIN_LN)
    for (int n = tid; n < D; n += 256) { sG[n] = a.gamma[n]; sBt[n] = a.beta[n]; }
  __syncthreads();

  auto issue_piece = [&](int s, int i) __attribute__((always_inline)) {      // stage s = (chunk s / SA, k stage s % SA)
    const int c = s / SA, r = s - c * SA;
    const char* src = W + ((size_t)(4 * c + w) * KC + 8 * r) * 512 + lane * 16;
    char* dst = sW + (s & (R - 1)) * RL_STAGE + w * 4096;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int s0 = 0; s0 < R - 1; ++s0)
    if (s0 < NS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) issue_piece(s0, i);
    }

  if constexpr (MODE == ROWLIN_LN) {
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NXF; ++i) sm += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
    sm += __shfl_xor(sm, 32, 64);
    const float mean = sm * (1.0f / D);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NXF; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = xv[i][e] - mean; ss += d * d; }
    ss += __shfl_xor(ss, 32, 64);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + a.eps);
#pragma unroll
    for (int t = 0; t < NXF; ++t) {
      u32x2 pk[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = 4 * t + 2 * half + j;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(sG + c * 4);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(sBt + c * 4);
        const f32x4 v = xv[2 * t + j];
        pk[j] = pack4<E>((v[0] - mean) * rstd * gm[0] + bt[0], (v[1] - mean) * rstd * gm[1] + bt[1],
                         (v[2] - mean) * rstd * gm[2] + bt[2], (v[3] - mean) * rstd * gm[3] + bt[3]);
      }
      const u32x4 q = {pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
      xf[t] = __builtin_bit_cast(V8, q);
    }
  }

  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int wo = half * 512 + r31 * 16;
  int s = 0;
  struct WF { V8 w[4]; };
  WF wf;
  auto stage_mid = [&](auto STEADY) __attribute__((always_inline)) {
    if constexpr (decltype(STEADY)::value) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * 4) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  constexpr int PRE = NS < R - 1 ? NS : R - 1;           // stages requested by the prologue
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PRE - 1) * 4) : "memory");       // stage 0 (own pieces) ...
  __builtin_amdgcn_s_barrier();                          // ... and everybody's
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 4; ++i) wf.w[i] = *reinterpret_cast<const V8*>(sW + wo + (i * 8) * 512);

  // parked chunk (MODE_LN): 8 units x 8 values, bias added, packed; unit u = tile u>>1, registers 8(u&1)..+7
  u32x4 pk[MODE == ROWLIN_LN ? 8 : 1];
  int pc = 0;                                            // chunk the parked values belong to
  char* ob = MODE == ROWLIN_LN ? static_cast<char*>(a.out) + rb * (N / 8) * 512 + r31 * 16 + half * 8 : nullptr;
  auto store_unit = [&](auto U) __attribute__((always_inline)) {             // two 8-byte stores: chunks 16*pc + 2u, +1
    constexpr int u = decltype(U)::value;
    const u32x2 lo = {pk[u][0], pk[u][1]}, hi = {pk[u][2], pk[u][3]};   // rows >= M of the last panel: padding rows, harmless
    *reinterpret_cast<u32x2*>(ob + (size_t)(16 * pc + 2 * u) * 512) = lo;
    *reinterpret_cast<u32x2*>(ob + (size_t)(16 * pc + 2 * u + 1) * 512) = hi;
  };
  f32x4 rv[MODE == ROWLIN_RESID ? 16 : 1];               // residual rows of the current chunk
  char* xr = MODE == ROWLIN_RESID ? reinterpret_cast<char*>(a.x) + rbc * (N / 4) * 512 + half * 512 + r31 * 16 : nullptr;

  // One chunk: SA stages x 4 k16 steps x 4 tiles.  REM0 = ring stages that follow the chunk (compile time, or FAR).
  constexpr int FAR = 1 << 20;
  auto chunk = [&](int c, auto AFTER, auto WITH_STORE) __attribute__((always_inline)) {
    constexpr bool with_store = decltype(WITH_STORE)::value;
    if constexpr (MODE == ROWLIN_RESID) {
#pragma unroll
      for (int g = 0; g < 16; ++g) rv[g] = *reinterpret_cast<const f32x4*>(xr + (size_t)(32 * c + 2 * g) * 512);
    }
    rfor<0, SA>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      constexpr int rem = decltype(AFTER)::value >= FAR ? FAR : decltype(AFTER)::value + (SA - 1 - ks);
      constexpr bool more = rem >= R - 1, next = rem >= 1;
      const char* st = sW + (s & (R - 1)) * RL_STAGE;
      const char* stn = sW + ((s + 1) & (R - 1)) * RL_STAGE;
      rfor<0, 4>([&](auto C4) {
        constexpr int c4 = decltype(C4)::value;
        if constexpr (c4 == 2) stage_mid(std::integral_constant<bool, (rem >= R - 2)>{});
        rfor<0, 4>([&](auto I) {
          constexpr int i = decltype(I)::value;
          acc[i] = Op16<E>::mfma(wf.w[i], xf[ks * 4 + c4], acc[i]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (c4 < 3) wf.w[i] = *reinterpret_cast<const V8*>(st + wo + (i * 8 + 2 * (c4 + 1)) * 512);
          else if constexpr (next) wf.w[i] = *reinterpret_cast<const V8*>(stn + wo + (i * 8) * 512);
          if constexpr (more && c4 >= 2 && i < 2) issue_piece(s + R - 1, (c4 - 2) * 2 + i);
          if constexpr (MODE == ROWLIN_LN && with_store) {
            constexpr int n = ks * 16 + c4 * 4 + i, NMM = SA * 16;           // unit k after MFMA ceil(k*NMM/8)
            constexpr int k = (n * 8) / NMM;
            if constexpr ((k * NMM + 7) / 8 == n && k < 8) store_unit(std::integral_constant<int, k>{});
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      });
      ++s;
    });
    // chunk done
    if constexpr (MODE == ROWLIN_LN) {
      rfor<0, 8>([&](auto U) {
        constexpr int u = decltype(U)::value, i = u >> 1, m = u & 1;
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sB + c * 128 + i * 32 + 8 * (2 * m) + 4 * half);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(sB + c * 128 + i * 32 + 8 * (2 * m + 1) + 4 * half);
        const u32x2 lo = pack4<E>(acc[i][8 * m] + b0[0], acc[i][8 * m + 1] + b0[1], acc[i][8 * m + 2] + b0[2], acc[i][8 * m + 3] + b0[3]);
        const u32x2 hi = pack4<E>(acc[i][8 * m + 4] + b1[0], acc[i][8 * m + 5] + b1[1], acc[i][8 * m + 6] + b1[2], acc[i][8 * m + 7] + b1[3]);
        const u32x4 p = {lo[0], lo[1], hi[0], hi[1]};
        pk[u] = p;
#pragma unroll
        for (int r = 8 * m; r < 8 * m + 8; ++r) acc[i][r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
      });
      pc = c;
    } else {
      rfor<0, 16>([&](auto G) {                          // g = 4*tile + q: features 128c + 32i + 8q + 4half ..+3
        constexpr int g = decltype(G)::value, i = g >> 2, q = g & 3;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(sB + c * 128 + i * 32 + 8 * q + 4 * half);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = acc[i][4 * q + e] + bv[e] + rv[g][e]; acc[i][4 * q + e] = 0.f; }
        *reinterpret_cast<f32x4*>(xr + (size_t)(32 * c + 2 * g) * 512) = o;
      });
    }
  };

  // chunks whose stages all see "plenty follows" run in a rolled loop; the trailing ones get their exact counts
  constexpr int TAILC = (R - 1 + SA - 1) / SA + 1 < NC ? (R - 1 + SA - 1) / SA + 1 : NC;
  typedef std::integral_constant<int, FAR> Far;
  if constexpr (NC > TAILC) {
    chunk(0, Far{}, std::false_type{});
#pragma unroll 1
    for (int c = 1; c < NC - TAILC; ++c) chunk(c, Far{}, std::true_type{});
  }
  rfor<0, TAILC>([&](auto T_) {
    constexpr int c = NC - TAILC + decltype(T_)::value;
    if constexpr (c == 0) chunk(c, std::integral_constant<int, (NC - 1 - c) * SA>{}, std::false_type{});
    else chunk(c, std::integral_constant<int, (NC - 1 - c) * SA>{}, std::true_type{});
  });
  if constexpr (MODE == ROWLIN_LN) rfor<0, 8>([&](auto U) { store_unit(U); });
}

template <typename E, int D, int N>
int launch_rl(int mode, const RowLinArgs& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.M + 127) / 128)), blk(256);
  if (mode == ROWLIN_LN) hipLaunchKernelGGL((rowlin_kernel<E, D, N, ROWLIN_LN>), grid, blk, 0, s, a);
  else hipLaunchKernelGGL((rowlin_kernel<E, D, N, ROWLIN_RESID>), grid, blk, 0, s, a);
  return check_launch("rowlin");
}

template <typename E>
int dispatch_rl(int mode, const RowLinArgs& a, hipStream_t s) {
  if (a.D == 384 && a.N == 1152) return launch_rl<E, 384, 1152>(mode, a, s);
  if (a.D == 384 && a.N == 384) return launch_rl<E, 384, 384>(mode, a, s);
  if (a.D == 128 && a.N == 384) return launch_rl<E, 128, 384>(mode, a, s);
  if (a.D == 128 && a.N == 128) return launch_rl<E, 128, 128>(mode, a, s);
  return fail(EFFOCR_EUNSUPPORTED, "rowlin: (D, N) must be (384, 1152|384) or (128, 384|128)");
}

}  // namespace

bool rowlin_supported(int prec, int D, int N) {
  return (prec == PREC_BF16 || prec == PREC_FP16) && ((D == 384 && (N == 1152 || N == 384)) || (D == 128 && (N == 384 || N == 128)));
}

int rowlin(int prec, int mode, const RowLinArgs& a, hipStream_t s) {
  if (a.M <= 0) return EFFOCR_OK;
  if (!rowlin_supported(prec, a.D, a.N)) return fail(EFFOCR_EUNSUPPORTED, "rowlin: needs bf16/fp16 and a supported (D, N)");
  if (mode == ROWLIN_RESID && a.N != a.D) return fail(EFFOCR_EINVAL, "rowlin: the residual mode needs N == D");
  if (a.rows_alloc % 128 != 0 || a.rows_alloc < a.M) return fail(EFFOCR_EINVAL, "rowlin: rows_alloc must be a multiple of 128 covering M (padding rows are written)");
  return prec == PREC_BF16 ? dispatch_rl<__bf16>(mode, a, s) : dispatch_rl<_Float16>(mode, a, s);
}

}  // namespace effocr
