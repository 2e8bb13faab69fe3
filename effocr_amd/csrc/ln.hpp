// Row LayerNorm helper shared by the standalone LayerNorm kernels (vit_ops.hip) and the LN-fused
// prologue of the row-panel GEMM (panel.hip).
#pragma once
#include "common.hpp"

namespace effocr {

// ------------------------------------------------------------------------------------------
// LayerNorm over rows of D fp32 values.  A row is owned by G lanes, V float4 per lane
// (D = 4*G*V), so a wave64 handles 64/G rows with 16-byte coalesced loads; statistics in fp32,
// two-pass (mean, then centred variance) like torch's CPU kernel.
// ------------------------------------------------------------------------------------------
template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LayerNorm of one row already in registers: v[i] = the lane's float4 at column (sub + G*i)*4
template <int G, int V>
__device__ __forceinline__ void ln_apply(const f32x4 (&v)[V], int sub, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, float eps, f32x4 (&y)[V]) {
  constexpr int D = 4 * G * V;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = group_sum<G>(s) * (1.0f / D);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; ss += d * d; }
  }
  const float var = group_sum<G>(ss) * (1.0f / D);
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + (sub + G * i) * 4);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + (sub + G * i) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[i][e] = (v[i][e] - mean) * rstd * gm[e] + bt[e];
  }
}

template <int G, int V>
__device__ __forceinline__ void ln_row(const float* __restrict__ xr, int sub, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float eps, f32x4 (&y)[V]) {
  f32x4 v[V];
#pragma unroll
  for (int i = 0; i < V; ++i) v[i] = *reinterpret_cast<const f32x4*>(xr + (sub + G * i) * 4);
  ln_apply<G, V>(v, sub, gamma, beta, eps, y);
}

}  // namespace effocr
