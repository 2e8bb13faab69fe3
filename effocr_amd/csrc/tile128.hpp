// 128x128 x (128-byte K-stage) register-staged MFMA tile shared by the NT GEMM (gemm.hip) and the
// inner-product k-NN kernel (knn.hip).  See gemm.hip for the structure notes.
#pragma once
#include "common.hpp"

namespace effocr {
namespace tile128 {

constexpr int BM = 128;          // tokens per tile
constexpr int BN = 128;          // features per tile
constexpr int ROWB = 128;        // payload bytes per row per K-stage
constexpr int ROWS = 144;        // padded LDS row stride (bytes)
constexpr int TILEB = 128 * ROWS;            // one operand, one stage
constexpr int STAGEB = 2 * TILEB;            // W + X
constexpr int GEMM_LDS = 2 * STAGEB;         // double buffered = 73,728 B

template <typename TA> struct is_f32 { static constexpr bool value = false; };
template <> struct is_f32<float> { static constexpr bool value = true; };

// ---- global -> register staging: 4 x 16 B per thread per operand
template <typename TA>
__device__ __forceinline__ void stage_load(u32x4 (&r)[4], const TA* __restrict__ base, int64_t ld,
                                           int row0, int nrows, int kbyte, int tid) {
  const int c = tid & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = row0 + (tid >> 3) + 32 * i;
    row = row < nrows ? row : nrows - 1;                 // clamp: tail rows are never stored
    const char* p = reinterpret_cast<const char*>(base + (int64_t)row * ld) + kbyte + c * 16;
    r[i] = *reinterpret_cast<const u32x4*>(p);
  }
}

template <typename TA>
__device__ __forceinline__ void stage_store(const u32x4 (&r)[4], char* tile, int tid) {
  const int c = tid & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    char* rowp = tile + ((tid >> 3) + 32 * i) * ROWS;
    if constexpr (is_f32<TA>::value) {
      // parity planes: even k -> bytes [0,64), odd k -> bytes [64,128)
      u32x2 ev = {r[i][0], r[i][2]};
      u32x2 od = {r[i][1], r[i][3]};
      *reinterpret_cast<u32x2*>(rowp + c * 8) = ev;
      *reinterpret_cast<u32x2*>(rowp + 64 + c * 8) = od;
    } else {
      *reinterpret_cast<u32x4*>(rowp + c * 16) = r[i];
    }
  }
}

// ---- one K-stage of MFMAs for this wave's 64x64 sub-tile
template <typename TA>
__device__ __forceinline__ void stage_mma(f32x16 (&acc)[2][2], const char* sW, const char* sX,
                                          int wn, int wm, int lane) {
  const int r31 = lane & 31, half = lane >> 5;
  const char* pw = sW + (wn * 64 + r31) * ROWS;
  const char* px = sX + (wm * 64 + r31) * ROWS;
  if constexpr (is_f32<TA>::value) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const f32x4*>(pw + i * 32 * ROWS + half * 64 + g * 16);
        b[i] = *reinterpret_cast<const f32x4*>(px + i * 32 * ROWS + half * 64 + g * 16);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
  } else {
    typedef typename Op16<TA>::V8 V8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      V8 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const V8*>(pw + i * 32 * ROWS + (2 * ks + half) * 16);
        b[i] = *reinterpret_cast<const V8*>(px + i * 32 * ROWS + (2 * ks + half) * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Op16<TA>::mfma(a[i], b[j], acc[i][j]);
    }
  }
}

}  // namespace tile128
}  // namespace effocr
