// resnet18 recognizer encoder (timm resnet18, num_classes=0 -> global-average-pooled 512-d feature;
// models/encoders.py:58, called at infer_effocr.py:314) — BASELINE.json config 1.
//
// Convolutions run as implicit GEMM on the shared fp32 MFMA tile pipeline (tile128.hpp,
// v_mfma_f32_32x32x2_f32): rows = output pixels (b,oy,ox), columns = output channels,
// K = (ky,kx,ci) with ci fastest, activations NHWC so that one 128-byte K-stage of a row is 32
// contiguous input channels of one tap -> the im2col gather is fused into the stage loader
// (zero-filled taps outside the image).  BatchNorm is folded into the weights/bias on the host
// (api.hip pack_resnet); bias, the residual add and ReLU are fused into the epilogue.
// conv1 (3 input channels, 7x7) has no 32-channel runs: a small im2col kernel builds its rows
// [B*OH*OW][160] straight from the NCHW input and the same kernel then runs as a 1x1 conv.
#include "common.hpp"
#include "kernels.hpp"
#include "tile128.hpp"

namespace effocr {
namespace {

using namespace tile128;

// a tile row's output pixel: top-left input coordinate of its window and the element index of (b, iy0, ix0, in_off + 4 c) — the index may
// lie outside the image (padding); it is only dereferenced for taps inside
struct RowCoord { int b, iy0, ix0, base; };
// the K-stage's tap (uniform over the workgroup): stage ks covers input channels ci0 .. ci0 + 32 of tap (ky, kx); advanced incrementally
// (round 4: the two integer divisions per stage, the 64-bit index arithmetic and the per-row bounds BRANCHES were 1 360 of a stage's 6 500
// cycles in front of its first MFMA — tools/conv_timeline.py)
struct TapState { int ky, kx, ci0; };
__device__ __forceinline__ TapState tap_of_stage(const ConvArgs& a, int ks) {
  const int kk = ks * 32, tap = kk / a.Cin;
  TapState t;
  t.ci0 = kk - tap * a.Cin; t.ky = tap / a.KW; t.kx = tap - t.ky * a.KW;
  return t;
}
__device__ __forceinline__ void tap_advance(TapState& t, const ConvArgs& a) {
  t.ci0 += 32;
  const bool wc = t.ci0 >= a.Cin;
  t.ci0 = wc ? 0 : t.ci0;
  t.kx += wc ? 1 : 0;
  const bool wx = t.kx >= a.KW;
  t.kx = wx ? 0 : t.kx;
  t.ky += wx ? 1 : 0;
}

// branch-free: a tap outside the image loads the tensor's first 16 bytes and is zeroed (conv_stage_zero, called behind the stage's MFMAs
// when the data is stored to LDS), so that the whole loop body is ONE basic block and the address arithmetic sits between the MFMAs
__device__ __forceinline__ void conv_stage_load(u32x4 (&r)[4], int (&ok)[4], const ConvArgs& a, const RowCoord (&rc)[4], const TapState& t) {
  const int off = (t.ky * a.W + t.kx) * a.in_ld + t.ci0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ok[i] = (int)((unsigned)(rc[i].iy0 + t.ky) < (unsigned)a.H) & (int)((unsigned)(rc[i].ix0 + t.kx) < (unsigned)a.W);   // & not &&: no control flow
    const int idx = ok[i] ? rc[i].base + off : 0;
    r[i] = *reinterpret_cast<const u32x4*>(a.in + idx);
  }
}
__device__ __forceinline__ void conv_stage_zero(u32x4 (&r)[4], const int (&ok)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = ok[i] ? r[i] : u32x4{0u, 0u, 0u, 0u};
}

// bf16-operand variant of the stage loader: a 128-byte stage row = 64 k values; chunk c (8 values = 16 bytes) is 8 consecutive
// input channels of ONE tap (Cin % 8 == 0), read as 8 fp32 and rounded to bf16 on the way (k >= K: the zero padding of the
// weight rows' last stage)
__device__ __forceinline__ void conv_stage_load16(u32x4 (&r)[4], const ConvArgs& a, const RowCoord (&rc)[4], int ks, int K, int tid) {
  const int k0 = ks * 64 + (tid & 7) * 8;
  const int tap = k0 / a.Cin, ci0 = k0 - tap * a.Cin;
  const int ky = tap / a.KW, kx = tap - ky * a.KW;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int iy = rc[i].iy0 + ky, ix = rc[i].ix0 + kx;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (k0 < K && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
      const float* p = a.in + (((int64_t)rc[i].b * a.H + iy) * a.W + ix) * a.in_ld + a.in_off + ci0;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
      const u32x2 l2 = pack4<__bf16>(lo[0], lo[1], lo[2], lo[3]), h2 = pack4<__bf16>(hi[0], hi[1], hi[2], hi[3]);
      v = u32x4{l2[0], l2[1], h2[0], h2[1]};
    }
    r[i] = v;
  }
}

// one K-stage of MFMAs for a wave's NI x NJ sub-tiles (32 channels x 32 pixels each): tile128::stage_mma with the wave's first weight / pixel
// row explicit, so that the channel tile can be narrower than 128 (NW below)
template <typename TA, int NI, int NJ>
__device__ __forceinline__ void conv_stage_mma(f32x16 (&acc)[NI][NJ], const char* sW, const char* sX, int wrow0, int xrow0, int lane) {
  const int r31 = lane & 31, half = lane >> 5;
  const char* pw = sW + (wrow0 + r31) * ROWS;
  const char* px = sX + (xrow0 + r31) * ROWS;
  if constexpr (tile128::is_f32<TA>::value) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a[NI], b[NJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) a[i] = *reinterpret_cast<const f32x4*>(pw + i * 32 * ROWS + half * 64 + g * 16);
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const f32x4*>(px + j * 32 * ROWS + half * 64 + g * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
      // the global loads the caller issued in front of this stage stay inside the first quarter of the MFMA stream (left alone, the
      // scheduler sinks them towards their use at the end of the stage and exposes their latency); LDS reads may cross the fence
      if (g == 0) __builtin_amdgcn_sched_barrier(0x100);
    }
  } else {
    typedef typename Op16<TA>::V8 V8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      V8 a[NI], b[NJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) a[i] = *reinterpret_cast<const V8*>(pw + i * 32 * ROWS + (2 * ks + half) * 16);
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const V8*>(px + j * 32 * ROWS + (2 * ks + half) * 16);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = Op16<TA>::mfma(a[i], b[j], acc[i][j]);
    }
  }
}

// -DCONV_STAMP (tools/ab_build.sh variant, never shipped): the waves of the first 256 workgroups of the 3x3 128 -> 128 layers record
// s_memtime at their per-stage milestones; tools/conv_timeline.py reads the last such launch's table through effocr_debug_conv_stamps.
#ifdef CONV_STAMP
constexpr int CONV_STAMP_WGS = 256, CONV_STAMP_N = 160;
__device__ unsigned long long conv_stamps[CONV_STAMP_WGS * 4 * CONV_STAMP_N];
#define CONV_STAMP_AT(k) if (stamp_on && lane == 0 && (k) < CONV_STAMP_N) conv_stamps[((int)blockIdx.x * 4 + w) * CONV_STAMP_N + (k)] = __builtin_amdgcn_s_memtime();
#else
#define CONV_STAMP_AT(k)
#endif
// E = float: fp32 operands on v_mfma_f32_32x32x2_f32 (exact products).  E = __bf16: operands rounded to bf16 (activations in the
// stage loader, weights once at upload: a.w16 [Cout][ceil(K/64)*64]) on v_mfma_f32_32x32x16_bf16, fp32 accumulation and epilogue.
// NW = channel tile (round 4): 128 (2 x 2 waves of 64 channels x 64 pixels), 64 (2 x 2 waves of 32 x 64) or 32 (4 waves of 32 x 32, all
// on the same 32 channels).  YOLOv5s spends 40 % of a forward in convolutions with <= 64 output channels (tools/loc_trace.py: C3 hidden
// widths 32 / 64, the first down-sampling convolutions): on the 128-wide tile 1/2 .. 3/4 of their MFMAs multiplied clamped weight rows.
template <typename E, int NW>
__global__ __launch_bounds__(256, (NW == 32 ? 3 : 2)) void conv_igemm_kernel(ConvArgs a) {
  constexpr bool B16 = !tile128::is_f32<E>::value;
  constexpr int NI = NW == 128 ? 2 : 1, NJ = NW == 32 ? 1 : 2;
  // a stage = [NW weight rows | 128 pixel rows] of 144 bytes, double buffered: 73.7 / 55.3 / 46.1 KB -> 2 / 2 / 3 workgroups per CU (the
  // narrow layers are the short-K, latency-bound ones: 1x1 convolutions of two to four stages)
  constexpr int WTB = NW * ROWS, STB = WTB + TILEB;
  __shared__ __attribute__((aligned(16))) char smem[2 * STB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = wave_id(), wn = w >> 1, wm = w & 1;
  const int M = a.B * a.OH * a.OW;
#ifdef CONV_STAMP
#ifdef CONV_STAMP_K1                                        // the 1x1 128 -> 128 layers instead (four K-stages)
  const bool stamp_on = NW == 128 && a.KH == 1 && a.Cin == 128 && a.Cout == 128 && a.ksplit <= 1 && blockIdx.x < CONV_STAMP_WGS;
#else
  const bool stamp_on = NW == 128 && a.KH == 3 && a.Cin == 128 && a.Cout == 128 && a.ksplit <= 1 && blockIdx.x < CONV_STAMP_WGS;
#endif
  CONV_STAMP_AT(0)
#endif
  const int K = a.KH * a.KW * a.Cin;
  const int ntn = (a.Cout + NW - 1) / NW;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * NW;
  const int wrow0 = NW == 128 ? wn * 64 : (NW == 64 ? wn * 32 : 0);      // this wave's first channel / pixel row inside the tile
  const int xrow0 = NW == 32 ? w * 32 : wm * 64;
  const int nks_all = B16 ? (K + 63) / 64 : K / 32;
  const int Kp = nks_all * 64;                           // (bf16) padded weight row length
  // split K: workgroup (x, y) runs the stages [y, y + 1) * nks_all / ksplit and writes its raw accumulators to the scratch
  const int ksp = a.ksplit > 1 ? a.ksplit : 1, sp = ksp > 1 ? (int)blockIdx.y : 0;
  const int ks_lo = (int)((int64_t)nks_all * sp / ksp), ks_hi = (int)((int64_t)nks_all * (sp + 1) / ksp);

  RowCoord rc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + (tid >> 3) + 32 * i;
    m = m < M ? m : M - 1;
    const int ox = m % a.OW, t = m / a.OW;
    const int oy = t % a.OH;
    rc[i].b = t / a.OH;
    rc[i].iy0 = oy * a.stride - a.pad;
    rc[i].ix0 = ox * a.stride - a.pad;
    rc[i].base = ((rc[i].b * a.H + rc[i].iy0) * a.W + rc[i].ix0) * a.in_ld + a.in_off + (tid & 7) * 4;   // < 2^31 elements: conv2d_nhwc checks
  }

  f32x16 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // weight rows of the tile: 4 x 32 per stage on the 128-wide tile; the narrow tiles stage only their NW rows (threads of the first
  // NW / 32 quarter passes: thread t copies row (t >> 3) + 32 i)
  constexpr int WI = NW / 32;
  u32x4 rw[WI], rx[4];
  unsigned woff[WI];                                         // byte offset of (this thread's weight row i, chunk c) in the weight matrix (< 4 GB)
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    int row = n0 + (tid >> 3) + 32 * i;
    row = row < a.Cout ? row : a.Cout - 1;                   // clamp: channels past Cout are never stored
    woff[i] = (unsigned)row * (unsigned)(B16 ? Kp * 2 : K * 4) + (tid & 7) * 16;
  }
  const char* wbase = B16 ? reinterpret_cast<const char*>(a.w16) : reinterpret_cast<const char*>(a.w);
  auto load_w = [&](int ks, u32x4 (&rw)[WI]) __attribute__((always_inline)) {
    const char* wb = wbase + (int64_t)ks * ROWB;             // uniform base + per-thread 32-bit offset
#pragma unroll
    for (int i = 0; i < WI; ++i) rw[i] = *reinterpret_cast<const u32x4*>(wb + woff[i]);
  };
  auto store_w = [&](char* tile, const u32x4 (&rw)[WI]) __attribute__((always_inline)) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      char* rowp = tile + ((tid >> 3) + 32 * i) * ROWS;
      if constexpr (!B16) {                                  // parity planes (tile128::stage_store): even k -> bytes [0,64), odd k -> [64,128)
        *reinterpret_cast<u32x2*>(rowp + c * 8) = u32x2{rw[i][0], rw[i][2]};
        *reinterpret_cast<u32x2*>(rowp + 64 + c * 8) = u32x2{rw[i][1], rw[i][3]};
      } else {
        *reinterpret_cast<u32x4*>(rowp + c * 16) = rw[i];
      }
    }
  };
  TapState tap = tap_of_stage(a, ks_lo);
  int okx[4] = {1, 1, 1, 1};
  auto load_stage = [&](int ks, u32x4 (&rw)[WI], u32x4 (&rx)[4]) __attribute__((always_inline)) {
    load_w(ks, rw);
    if constexpr (B16) conv_stage_load16(rx, a, rc, ks, K, tid);
    else conv_stage_load(rx, okx, a, rc, tap);
  };
  load_stage(ks_lo, rw, rx);
  if constexpr (!B16) conv_stage_zero(rx, okx);
  store_w(smem, rw);
  stage_store<E>(rx, smem + WTB, tid);
  __syncthreads();
  CONV_STAMP_AT(1)
  if constexpr (B16) {
    for (int ks = ks_lo; ks < ks_hi; ++ks) {
      char* cur = smem + ((ks - ks_lo) & 1) * STB;
      char* nxt = smem + (((ks - ks_lo) & 1) ^ 1) * STB;
      const bool more = (ks + 1) < ks_hi;
      if (more) load_stage(ks + 1, rw, rx);
      conv_stage_mma<E, NI, NJ>(acc, cur, cur + WTB, wrow0, xrow0, lane);
      if (more) {
        store_w(nxt, rw);
        stage_store<E>(rx, nxt + WTB, tid);
      }
      __syncthreads();
    }
  } else {
    // One basic block per stage: the next stage's loads are requested inside the first quarter of this stage's MFMAs (fence in
    // conv_stage_mma), zeroed / stored to the idle buffer behind the last MFMA; the last stage re-loads itself (stored, never read).
    // (Loads TWO stages ahead with the LDS stores inside the MFMA stream: +32..88 registers, 3.98 vs 3.95 ms per 16-image forward.)
    for (int ks = ks_lo; ks < ks_hi; ++ks) {
      char* cur = smem + ((ks - ks_lo) & 1) * STB;
      char* nxt = smem + (((ks - ks_lo) & 1) ^ 1) * STB;
      if (ks + 1 < ks_hi) tap_advance(tap, a);
      load_stage(ks + 1 < ks_hi ? ks + 1 : ks, rw, rx);
      CONV_STAMP_AT(4 + 4 * (ks - ks_lo))
      conv_stage_mma<E, NI, NJ>(acc, cur, cur + WTB, wrow0, xrow0, lane);
      CONV_STAMP_AT(5 + 4 * (ks - ks_lo))
      __builtin_amdgcn_sched_barrier(0);                  // nothing that consumes the loads moves up into the MFMA stream
      conv_stage_zero(rx, okx);
      store_w(nxt, rw);
      stage_store<E>(rx, nxt + WTB, tid);
      CONV_STAMP_AT(6 + 4 * (ks - ks_lo))
      __syncthreads();
      CONV_STAMP_AT(7 + 4 * (ks - ks_lo))
    }
  }

  const int half = lane >> 5;
  if (ksp > 1) {                                         // raw partial sums [split][M][Cout]; conv_reduce_kernel finishes (direct stores: through
#pragma unroll                                           // the LDS tile like the epilogue below measured 0.4 % slower — the deep layers' tiles are few)
    for (int j = 0; j < NJ; ++j) {
      const int m = m0 + xrow0 + j * 32 + (lane & 31);
      if (m >= M) continue;
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wrow0 + i * 32 + 8 * q + 4 * half;
          if (n >= a.Cout) continue;
          const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          *reinterpret_cast<f32x4*>(a.partial + ((int64_t)sp * M + m) * a.Cout + n) = v;
        }
    }
    return;
  }
  // Epilogue through LDS (round 4).  In the MFMA result layout a lane holds 4 channels of ONE pixel: a `global_store_dwordx4` of the wave is 32
  // separate 32-byte pieces (230 cycles per instruction on the stamps, 16 of them per lane) and the residual read has the same shape.  The
  // stage buffers are free behind the last barrier: bias + SiLU in the result layout, tile to LDS as [pixel][NW + 4 channels], then every wave
  // instruction moves whole pixel rows (NW = 128: two rows of 512 bytes) — residual read, ReLU and the store in that layout.
  constexpr int TS = NW + 4;                               // floats per pixel row of the transposed tile (odd multiple of 4: conflict-free)
  static_assert(128 * TS * 4 <= 2 * STB, "conv epilogue: the output tile must fit the stage buffers");
  float* tile = reinterpret_cast<float*>(smem);
  const int cmax = a.Cout - 4;                            // Cout % 4 == 0
  f32x4 bv[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + wrow0 + i * 32 + 8 * q + 4 * half;
      bv[i][q] = *reinterpret_cast<const f32x4*>(a.bias + (n < cmax ? n : cmax));
    }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        v += bv[i][q];
        if (a.silu) {                           // x * sigmoid(x), before the residual (Bottleneck: x + cv2(cv1(x)))
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_fast(v[e]);
        }
        *reinterpret_cast<f32x4*>(tile + (xrow0 + j * 32 + (lane & 31)) * TS + wrow0 + i * 32 + 8 * q + 4 * half) = v;
      }
  __syncthreads();
  constexpr int C4 = NW / 4;                               // 16-byte pieces per pixel row
#pragma unroll
  for (int it = 0; it < 128 * C4 / 256; ++it) {
    const int idx = it * 256 + tid, p = idx / C4, c4 = idx - p * C4;
    const int m = m0 + p, n = n0 + c4 * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(tile + p * TS + c4 * 4);
    const bool ok = m < M && n < a.Cout;
    if (a.resid && ok) v += *reinterpret_cast<const f32x4*>(a.resid + (int64_t)m * a.res_ld + a.res_off + n);
    if (a.relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (ok) *reinterpret_cast<f32x4*>(a.out + (int64_t)m * a.out_ld + a.out_off + n) = v;
  }
  CONV_STAMP_AT(2)
}

// split-K epilogue: out = act(sum over splits (fixed order) + bias) (+ residual) — one thread per (output pixel, 4 channels)
__global__ __launch_bounds__(256) void conv_reduce_kernel(ConvArgs a, int64_t M) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int nq = a.Cout / 4;
  if (id >= M * nq) return;
  const int64_t m = id / nq;
  const int n = (int)(id - m * nq) * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int sp = 0; sp < a.ksplit; ++sp) v += *reinterpret_cast<const f32x4*>(a.partial + ((int64_t)sp * M + m) * a.Cout + n);
  v += *reinterpret_cast<const f32x4*>(a.bias + n);
  if (a.silu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = silu_fast(v[e]);
  }
  if (a.resid) v += *reinterpret_cast<const f32x4*>(a.resid + m * a.res_ld + a.res_off + n);
  if (a.relu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  *reinterpret_cast<f32x4*>(a.out + m * a.out_ld + a.out_off + n) = v;
}

// conv1 im2col: col[(b,oy,ox)][(ky*7+kx)*3 + c] = x[b][c][2oy-3+ky][2ox-3+kx] (0 outside), cols 147..159 = 0
__global__ __launch_bounds__(256) void im2col_conv1_kernel(const float* __restrict__ x, float* __restrict__ col,
                                                           int B, int H, int W, int OH, int OW) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * OH * OW * 160;
  if (id >= total) return;
  const int k = (int)(id % 160);
  const int64_t m = id / 160;
  float v = 0.f;
  if (k < 147) {
    const int c = k % 3, tap = k / 3, kx = tap % 7, ky = tap / 7;
    const int ox = (int)(m % OW), oy = (int)((m / OW) % OH);
    const int64_t b = m / ((int64_t)OW * OH);
    const int iy = oy * 2 - 3 + ky, ix = ox * 2 - 3 + kx;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[((b * 3 + c) * H + iy) * (int64_t)W + ix];
  }
  col[id] = v;
}

// max_pool2d(kernel 3, stride 2, padding 1) on NHWC (padding never wins: -inf)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                      int B, int H, int W, int C, int OH, int OW) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int C4 = C / 4;
  const int64_t total = (int64_t)B * OH * OW * C4;
  if (id >= total) return;
  const int c4 = (int)(id % C4);
  const int64_t p = id / C4;
  const int ox = (int)(p % OW), oy = (int)((p / OW) % OH);
  const int64_t b = p / ((int64_t)OW * OH);
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((b * H + iy) * W + ix) * C + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    }
  *reinterpret_cast<f32x4*>(out + p * C + c4 * 4) = m;
}

// global average pool over HW (+ optional L2 normalisation): one workgroup per image, C = 512
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                      int HW, int C, int l2norm) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float v[2];
  float ss = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int c = tid + 256 * t;
    float s = 0.f;
    if (c < C) {
      for (int p = 0; p < HW; ++p) s += in[((int64_t)b * HW + p) * C + c];
      s = s / (float)HW;
    }
    v[t] = s;
    ss += s * s;
  }
  if (l2norm) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
    v[0] = v[0] / nrm; v[1] = v[1] / nrm;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int c = tid + 256 * t;
    if (c < C) out[(int64_t)b * C + c] = v[t];
  }
}

}  // namespace

int conv2d_nhwc(const ConvArgs& a_in, hipStream_t s) {
  ConvArgs a = a_in;
  if (a.in_ld == 0) a.in_ld = a.Cin;
  if (a.out_ld == 0) a.out_ld = a.Cout;
  if (a.res_ld == 0) a.res_ld = a.Cout;
  const int64_t M = (int64_t)a.B * a.OH * a.OW;
  if (M <= 0) return EFFOCR_OK;
  if ((a.in_ld | a.in_off | a.out_ld | a.out_off | a.res_ld | a.res_off) & 3) return fail(EFFOCR_EUNSUPPORTED, "conv2d: channel strides / offsets must be multiples of 4");
  if (a.Cin % 32 != 0 || a.Cout % 4 != 0) return fail(EFFOCR_EUNSUPPORTED, "conv2d: Cin must be a multiple of 32 and Cout of 4");
  if (M >= ((int64_t)1 << 31) - 256) return fail(EFFOCR_EUNSUPPORTED, "conv2d: too many output pixels");
  // the stage loader indexes the input with 32-bit element offsets and the weights with 32-bit byte offsets
  if ((int64_t)a.Cout * (((int64_t)a.KH * a.KW * a.Cin + 63) / 64 * 64) * 4 >= ((int64_t)1 << 32)) return fail(EFFOCR_EUNSUPPORTED, "conv2d: weights of 4 GB or more");
  if ((int64_t)a.B * a.H * a.W * a.in_ld >= ((int64_t)1 << 31)) {      // halve the batch until a launch's input fits
    if (a.B < 2) return fail(EFFOCR_EUNSUPPORTED, "conv2d: one image of 2^31 or more input elements");
    ConvArgs lo = a, hi = a;
    lo.B = a.B / 2; hi.B = a.B - lo.B;
    hi.in = a.in + (int64_t)lo.B * a.H * a.W * a.in_ld;
    hi.out = a.out + (int64_t)lo.B * a.OH * a.OW * a.out_ld;
    if (a.resid) hi.resid = a.resid + (int64_t)lo.B * a.OH * a.OW * a.res_ld;
    const int rc = conv2d_nhwc(lo, s);
    return rc ? rc : conv2d_nhwc(hi, s);
  }
  int nw = a.Cout <= 32 ? 32 : (a.Cout <= 64 ? 64 : 128);             // channel tile: narrow layers do not multiply clamped weight rows
  int64_t grid = ((M + BM - 1) / BM) * ((a.Cout + nw - 1) / nw);
  // between half a round and one round of 128-channel tiles (16 images: the 40 x 40 maps, 200 tiles on 256 CUs, ONE workgroup per CU with
  // nothing to run under its barriers): 64-channel tiles instead, two co-resident workgroups on most CUs (3.78 -> 3.72 ms per forward)
  if (nw == 128 && grid * 2 > device_cus() && grid <= device_cus()) { nw = 64; grid = ((M + BM - 1) / BM) * ((a.Cout + nw - 1) / nw); }
  // few tiles and a long K (the deep layers at small inputs: 4 workgroups looping over 144 stages): split K over up to a round of CUs
  const int nks = a.w16 ? (a.KH * a.KW * a.Cin + 63) / 64 : a.KH * a.KW * a.Cin / 32;
  a.ksplit = 1;
  const int cus = device_cus();
  if (a.partial && grid * 2 <= cus && nks >= 8) {
    int64_t sp = cus / grid;
    if (sp > nks / 2) sp = nks / 2;
    if (sp > 32) sp = 32;
    while (sp > 1 && (size_t)sp * M * a.Cout * 4 > a.partial_bytes) --sp;
    a.ksplit = (int)sp;
  }
  const dim3 g((unsigned)grid, (unsigned)a.ksplit);
  if (a.w16) {
    if (nw == 32) hipLaunchKernelGGL((conv_igemm_kernel<__bf16, 32>), g, dim3(256), 0, s, a);
    else if (nw == 64) hipLaunchKernelGGL((conv_igemm_kernel<__bf16, 64>), g, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_igemm_kernel<__bf16, 128>), g, dim3(256), 0, s, a);
  } else {
    if (nw == 32) hipLaunchKernelGGL((conv_igemm_kernel<float, 32>), g, dim3(256), 0, s, a);
    else if (nw == 64) hipLaunchKernelGGL((conv_igemm_kernel<float, 64>), g, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_igemm_kernel<float, 128>), g, dim3(256), 0, s, a);
  }
  int rc = check_launch("conv2d_nhwc");
  if (rc || a.ksplit == 1) return rc;
  const int64_t items = M * (a.Cout / 4);
  hipLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, a, M);
  return check_launch("conv_reduce");
}

int im2col_conv1(const float* x, float* col, int B, int H, int W, int OH, int OW, hipStream_t s) {
  const int64_t total = (int64_t)B * OH * OW * 160;
  if (total <= 0) return EFFOCR_OK;
  hipLaunchKernelGGL(im2col_conv1_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, col, B, H, W, OH, OW);
  return check_launch("im2col_conv1");
}

int maxpool3x3s2_nhwc(const float* in, float* out, int B, int H, int W, int C, int OH, int OW, hipStream_t s) {
  const int64_t total = (int64_t)B * OH * OW * (C / 4);
  if (total <= 0) return EFFOCR_OK;
  hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, B, H, W, C, OH, OW);
  return check_launch("maxpool");
}

int global_avgpool_nhwc(const float* in, float* out, int B, int HW, int C, int l2norm, hipStream_t s) {
  if (B <= 0) return EFFOCR_OK;
  if (C > 512) return fail(EFFOCR_EUNSUPPORTED, "avgpool: more than 512 channels");
  hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)B), dim3(256), 0, s, in, out, HW, C, l2norm);
  return check_launch("avgpool");
}

}  // namespace effocr

#ifdef CONV_STAMP
extern "C" int effocr_debug_conv_stamps(unsigned long long* out, int n) {
  const int m = effocr::CONV_STAMP_WGS * 4 * effocr::CONV_STAMP_N;
  if (n < m) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(effocr::conv_stamps), (size_t)m * 8) == hipSuccess ? 0 : -2;
}
#endif
