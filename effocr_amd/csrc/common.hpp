// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the EffOCR recognizer path.
// wave = 64 lanes everywhere; no portability layer — this code targets gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include "../../include/effocr_hip.h"

namespace effocr {

// ---------------------------------------------------------------- vector types / MFMA fragments
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 f16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t u32x4;
typedef __attribute__((__vector_size__(2 * sizeof(uint32_t)))) uint32_t u32x2;

struct bf16_t { typedef __bf16 T; typedef bf16x8 V8; };
struct f16_t  { typedef _Float16 T; typedef f16x8 V8; };

// 16-bit operand traits (bf16 / f16): storage type, 8-wide fragment, MFMA 32x32x16.
template <typename E> struct Op16;
template <> struct Op16<__bf16> {
  typedef bf16x8 V8;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Op16<_Float16> {
  typedef f16x8 V8;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// MFMA 32x32 C/D fragment map (dtype independent on gfx950): lane l, register r in [0,16):
//   col = l & 31,  row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// wave id as a provably wave-uniform scalar
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// XCD-aware block remap (8 XCDs, block b is dispatched to XCD b % 8): consecutive *logical* ids
// land on the same XCD so that neighbouring tiles share that XCD's L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, in = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + in;
}

// Fragment-blocked activation layout (fast path): a [rows, cols] matrix of 16-byte chunks is stored as
// cells [row/32][chunk][row%32][16 B] (512 B per cell).  In the swapped-MFMA C/D layout a lane owns 4
// consecutive features of token (lane&31), so one wave store / load instruction covers ONE contiguous
// cell per half-wave instead of 32 partial cache lines (78 -> <=37 TA cycles per instruction measured).
__host__ __device__ __forceinline__ int64_t blk_off(int64_t row, int chunk, int nchunks) {
  return ((row >> 5) * nchunks + chunk) * 512 + (row & 31) * 16;
}

// erf-based GELU (torch.nn.GELU default, what timm's Mlp uses): 0.5 x (1 + erf(x / sqrt 2))
// SiLU x * sigmoid(x) on the hardware transcendentals: v_exp_f32 (2^t) and v_rcp_f32, 1 ulp each -> <= 3e-7 relative to the
// expf / IEEE-division form (which costs 24 instructions per value: a quarter of a short convolution's tile time was its epilogue).
// x -> -inf: 2^t = inf, rcp = 0, x * 0 = -0 (as x / inf); x -> +inf: 2^t = 0 (denormals do not matter next to 1), result x.
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f)); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Same function for results that are rounded to bf16/f16 anyway (2^-9 / 2^-12 relative): erf from a
// degree-8 odd minimax fit erf(z) ~ z*P(z^2) on |z| <= 3 (|abs err| <= 2.3e-5, erf(3) = 0.99998), z
// clamped to +-3.  Transcendental-free and branch-free: 14 plain VALU per element — libm's
// two-branch erff (or an exp-based form) made the fc1 epilogue VALU-bound (rocprof: 14 VALU/MFMA).
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = __builtin_amdgcn_fmed3f(x * 0.70710678118654752440f, -3.0f, 3.0f);
  const float t = z * z;
  float p = fmaf(4.07419588e-08f, t, -1.94481757e-06f);
  p = fmaf(p, t, 4.1060451e-05f);
  p = fmaf(p, t, -0.00051103633f);
  p = fmaf(p, t, 0.00423542528f);
  p = fmaf(p, t, -0.0251028568f);
  p = fmaf(p, t, 0.111079332f);
  p = fmaf(p, t, -0.375314877f);
  p = fmaf(p, t, 1.12826843f);
  const float h = 0.5f * x;
  return fmaf(z * p, h, h);                              // 0.5x + 0.5x*erf(x/sqrt2)
}

// N-wide form: every Horner step is N independent FMAs, so the compiler can emit packed
// v_pk_fma_f32 without dependency nops (the scalar form serialised into fma -> s_nop -> fma chains)
template <int N> __device__ __forceinline__ void gelu_erf_fast_n(float (&x)[N]) {
  float z[N], t[N], p[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    z[i] = __builtin_amdgcn_fmed3f(x[i] * 0.70710678118654752440f, -3.0f, 3.0f);
    t[i] = z[i] * z[i];
    p[i] = fmaf(4.07419588e-08f, t[i], -1.94481757e-06f);
  }
  constexpr float c[7] = {4.1060451e-05f, -0.00051103633f, 0.00423542528f, -0.0251028568f, 0.111079332f, -0.375314877f, 1.12826843f};
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = fmaf(p[i], t[i], c[k]);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float h = 0.5f * x[i];
    x[i] = fmaf(z[i] * p[i], h, h);
  }
}

// GELU with the constants folded: gelu(x) = x * (0.5 + u*q(u^2)), u = clamp(x, +-L), u*q(u^2) ~ 0.5*erf(u/sqrt2); q = minimax fit of the
// weighted error |x| |Phi~(x) - Phi(x)| over |x| <= 12 (linear program on a dense grid, tools/fit_gelu.py — round 4 refit: the degree-6
// polynomial went from 3.9e-4 to 1.45e-4).  bf16 outputs (8-bit mantissa) take degree 6 (|gelu err| <= 1.5e-4, L = 3.9); f16 outputs
// GELU_DEG_F16: 6 (default), 7 (<= 6.9e-5, L = 4.1) or 8 (<= 6.4e-5, L = 4.2) — every degree is 4 VALU instructions per quad of values
// in the fused MLP's hand-over list, where ~6 issue slots per MFMA gap are free (mlp_kernel.hpp).  Same-box A/B (gpurun_out/f16ab*.txt):
// f16 step 11.43 / 11.35 / 11.22 ms at degree 8 / 7 / 6, embedding error 7.2-7.8e-4 at every degree (the f16 operand roundings dominate).
#ifndef GELU_DEG_F16
#define GELU_DEG_F16 6
#endif
template <typename E> struct GeluFit {
  static constexpr bool lo = sizeof(E) == 2 && !__is_same(E, _Float16);
  static constexpr int DEG = lo ? 6 : (sizeof(E) == 2 ? GELU_DEG_F16 : 8);
  static constexpr float L = DEG == 6 ? 3.9f : DEG == 7 ? 4.1f : 4.2f;
  static __device__ __forceinline__ constexpr float c(int k) {
    constexpr float c6[7] = {3.980856134e-01f, -6.486948046e-02f, 8.913788161e-03f, -8.445121550e-04f, 5.123729042e-05f, -1.770496053e-06f, 2.627150211e-08f};
    constexpr float c7[8] = {3.985286087e-01f, -6.563282613e-02f, 9.356159468e-03f, -9.656455460e-04f, 6.894064765e-05f, -3.190561829e-06f, 8.525671410e-08f, -9.917594109e-10f};
    constexpr float c8[9] = {3.989074382e-01f, -6.636037144e-02f, 9.830130026e-03f, -1.114147779e-03f, 9.457434347e-05f, -5.760762241e-06f, 2.343669162e-07f, -5.633299462e-09f, 5.998041464e-11f};
    return DEG == 6 ? c6[k < 7 ? k : 6] : DEG == 7 ? c7[k < 8 ? k : 7] : c8[k];
  }
};
template <typename E, int N> __device__ __forceinline__ void gelu_fold_n(float (&x)[N]) {
  typedef GeluFit<E> GF;
  constexpr float L = GF::L;
  constexpr int DEG = GF::DEG;
  if constexpr (N == 8 || N == 16) {
    constexpr int NCH = N / 2;                           // packed chains in lock step (8 of them: a dependent pair is 8 issues apart, no wait states at all)
    // Eight values = FOUR packed chains kept in lock step.  A v_pk_fma_f32 consuming the previous packed result needs a wait
    // state; left alone the scheduler walks one chain at a time to save registers (an s_nop behind every packed FMA: 277 per
    // 192 MFMAs in the fused MLP's loop), and a sched_barrier does not survive instruction selection — the empty asm with
    // every chain value as an in/out operand does.  Same arithmetic, value for value, as the scalar form below.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 u[NCH], t[NCH], p[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      u[i] = f32x2{__builtin_amdgcn_fmed3f(x[2 * i], -L, L), __builtin_amdgcn_fmed3f(x[2 * i + 1], -L, L)};
      t[i] = u[i] * u[i];
      const float ch = GF::c(DEG), cl = GF::c(DEG - 1);
      p[i] = __builtin_elementwise_fma(f32x2{ch, ch}, t[i], f32x2{cl, cl});
    }
    // join: every chain is at the same step before any moves on.  INPUT-only operands — an asm that (re)defines the chain registers
    // makes the hazard recognizer assume a forwarding hazard on them: two s_nop per step, 129 per 192 MFMAs in the fused MLP's loop.
    auto join = [&](const f32x2 (&q)[NCH]) __attribute__((always_inline)) {
      if constexpr (NCH == 4) asm volatile("" :: "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]));
      else asm volatile("" :: "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4 % NCH]), "v"(q[5 % NCH]), "v"(q[6 % NCH]), "v"(q[7 % NCH]));
    };
#pragma unroll
    for (int k = DEG - 2; k >= 0; --k) {
      join(p);
      const float ck = GF::c(k);
#pragma unroll
      for (int i = 0; i < NCH; ++i) p[i] = __builtin_elementwise_fma(p[i], t[i], f32x2{ck, ck});
    }
    join(p);
#pragma unroll
    for (int i = 0; i < NCH; ++i) p[i] = __builtin_elementwise_fma(u[i], p[i], f32x2{0.5f, 0.5f});
    join(p);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const f32x2 xv = {x[2 * i], x[2 * i + 1]};
      const f32x2 r = xv * p[i];
      x[2 * i] = r[0]; x[2 * i + 1] = r[1];
    }
    return;
  }
  float u[N], t[N], p[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    u[i] = __builtin_amdgcn_fmed3f(x[i], -L, L);
    t[i] = u[i] * u[i];
    p[i] = fmaf(GF::c(DEG), t[i], GF::c(DEG - 1));
  }
#pragma unroll
  for (int k = DEG - 2; k >= 0; --k)
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = fmaf(p[i], t[i], GF::c(k));
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = x[i] * fmaf(u[i], p[i], 0.5f);
}

// two floats -> one 32-bit word of two 16-bit values (first value in the low half), and back
template <typename E> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  typedef __attribute__((__vector_size__(2 * sizeof(E)))) E V2;
  typedef float F2 __attribute__((__vector_size__(8)));
  const F2 f = {a, b};
  const V2 v = __builtin_convertvector(f, V2);            // ONE v_cvt_pk_*: a vector conversion, not two scalar ones the vectoriser would have to find
  return __builtin_bit_cast(uint32_t, v);
}
template <typename E> __device__ __forceinline__ float unpack1(uint32_t wd, int hi) {
  if constexpr (GeluFit<E>::lo) return __builtin_bit_cast(float, hi ? (wd & 0xffff0000u) : (wd << 16));
  else {
    typedef __attribute__((__vector_size__(2 * sizeof(E)))) E V2;
    const V2 v = __builtin_bit_cast(V2, wd);
    return (float)v[hi];
  }
}

template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }
template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }

// pack 4 floats into 4 x 16-bit (8 bytes) / 2 floats
template <typename E> __device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
  typedef __attribute__((__vector_size__(4 * sizeof(E)))) E V4;
  V4 v = {(E)a, (E)b, (E)c, (E)d};
  return __builtin_bit_cast(u32x2, v);
}

// ---------------------------------------------------------------- host side error plumbing
// status codes: enum effocr_status in include/effocr_hip.h

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int check_launch(const char* what);
// CU count of the CURRENT device (hipGetDevice), cached per device id: launch geometry follows the device a call runs on,
// not the first device the process ever used (a process may drive several GPUs)
int device_cus();

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace effocr
