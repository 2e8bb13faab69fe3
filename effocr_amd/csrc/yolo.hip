// Non-convolution kernels of the YOLOv5 character / word localizer (BASELINE.json config 5; reference:
// onnx_engines/localizer_engine.py — an ONNXRuntime session over an ultralytics YOLOv5 export + letterbox + NMS).
// Activations are NHWC fp32 like the resnet path; every kernel reads / writes CHANNEL SLICES of wider buffers so that
// the network's concatenations (C3, SPPF, neck) are never copied.  The convolutions themselves are resnet.hip's
// implicit-GEMM kernel (exact fp32 MFMA) with a SiLU epilogue.
//   im2col_nchw     stem conv (3 input channels, 6x6 / stride 2): rows for the 1x1 implicit GEMM
//   upsample2x      nn.Upsample(scale_factor=2, mode="nearest")
//   maxpool5        nn.MaxPool2d(5, 1, 2) (SPPF)
//   yolo_decode     ultralytics Detect.forward in inference mode: sigmoid, grid / anchor decode, (bs, na*ny*nx, no) layout
//   letterbox_u8    EffLocalizer.letterbox + load_localizer_img (localizer_engine.py:75-138): cv2.resize(INTER_LINEAR)
//                   fixed-point bilinear, 114-grey border, BGR->RGB, /255, CHW
//   nms_*           EffLocalizer.non_max_suppression (localizer_engine.py:171-277): objectness / class-confidence filter,
//                   best class, confidence sort, class-offset boxes, torchvision.ops.nms, max_det
#include "common.hpp"
#include "kernels.hpp"
#include <math.h>

namespace effocr {
namespace {

__global__ __launch_bounds__(256) void im2col_nchw_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int Cin, int H, int W,
                                                          int KH, int KW, int stride, int pad, int OH, int OW, int kpad) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * OH * OW * kpad;
  if (id >= total) return;
  const int k = (int)(id % kpad);
  const int64_t m = id / kpad;
  float v = 0.f;
  if (k < KH * KW * Cin) {
    const int c = k % Cin, tap = k / Cin, kx = tap % KW, ky = tap / KW;
    const int ox = (int)(m % OW), oy = (int)((m / OW) % OH);
    const int64_t b = m / ((int64_t)OW * OH);
    const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[((b * Cin + c) * H + iy) * (int64_t)W + ix];
  }
  col[id] = v;
}

// YOLOv5 stem Conv(3, 32, 6, 2, 2) + folded BN + SiLU straight from the NCHW input: out NHWC [B,OH,OW,out_ld] channels out_off .. +32.
// The im2col detour wrote and re-read 128 fp32 per output pixel (1.7 GB per 16 letterboxed images: 0.88 ms + its GEMM); the layer is
// 108 taps x 32 channels per pixel, i.e. VALU work: a thread owns TWO horizontally adjacent pixels (their 6-wide windows share 4 of 8
// columns) x 32 channels = 64 accumulators, taps ascending (ky, kx, c) like the weight rows [co][(ky*6 + kx)*3 + c]; the weights sit
// transposed in LDS ([tap][co]: every lane reads the same 16 bytes — broadcast reads).
__global__ __launch_bounds__(256) void stem6x6s2_kernel(const float* __restrict__ x, const float* __restrict__ w, int w_ld, const float* __restrict__ bias,
                                                        float* __restrict__ out, int B, int H, int W, int OH, int OW, int out_ld, int out_off, int silu) {
  __shared__ __attribute__((aligned(16))) float wt[108 * 32];
  for (int i = threadIdx.x; i < 108 * 32; i += 256) { const int k = i >> 5, co = i & 31; wt[i] = w[(size_t)co * w_ld + k]; }
  __syncthreads();
  const int OW2 = (OW + 1) >> 1;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (int64_t)B * OH * OW2) return;
  const int ox = (int)(id % OW2) * 2, oy = (int)((id / OW2) % OH);
  const int64_t b = id / ((int64_t)OW2 * OH);
  float acc[2][32];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int co = 0; co < 32; ++co) acc[p][co] = 0.f;
  const int ix0 = ox * 2 - 2;
#pragma unroll 1
  for (int ky = 0; ky < 6; ++ky) {
    const int iy = oy * 2 - 2 + ky;
    const bool rowok = iy >= 0 && iy < H;
    float v[3][8];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* xr = x + ((b * 3 + c) * H + (rowok ? iy : 0)) * (int64_t)W;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int ix = ix0 + j; v[c][j] = (rowok && ix >= 0 && ix < W) ? xr[ix] : 0.f; }
    }
#pragma unroll
    for (int kx = 0; kx < 6; ++kx)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* wr = wt + ((ky * 6 + kx) * 3 + c) * 32;
        const float a0 = v[c][kx], a1 = v[c][kx + 2];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc[0][4 * q + e] = fmaf(a0, wv[e], acc[0][4 * q + e]); acc[1][4 * q + e] = fmaf(a1, wv[e], acc[1][4 * q + e]); }
        }
      }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (ox + p >= OW) break;
    float* o = out + ((b * OH + oy) * (int64_t)OW + ox + p) * out_ld + out_off;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * q);
      f32x4 r = {acc[p][4 * q] + bv[0], acc[p][4 * q + 1] + bv[1], acc[p][4 * q + 2] + bv[2], acc[p][4 * q + 3] + bv[3]};
      if (silu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = silu_fast(r[e]);
      }
      *reinterpret_cast<f32x4*>(o + 4 * q) = r;
    }
  }
}

// Round 4: the kernel above is bound by its LDS weight reads (one broadcast ds_read_b128 per 8 FMAs: 272 us per 16 images for 82 us of
// packed FMAs).  Here a thread owns FOUR horizontally adjacent pixels x 32 channels (128 accumulators; their windows share 8 of 12
// columns) and the weights never touch LDS: `wt` is the [tap][co] transpose in global memory, its addresses are uniform, so the compiler
// fetches them with s_load_dwordx16 through the scalar cache (13.8 KB, resident) and feeds them to v_pk_fma_f32 as SGPR pairs.  Inputs
// as 8-byte pairs (W even: a pair is inside or outside the row as a whole), branch-free.  Same tap order (ky, kx, c) and fmaf chain per
// output as above: bit-identical.
__global__ __launch_bounds__(256, 2) void stem6x6s2_sw_kernel(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                                                           float* __restrict__ out, int B, int H, int W, int OH, int OW, int out_ld, int out_off, int silu) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int OW4 = (OW + 3) >> 2;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (int64_t)B * OH * OW4) return;
  const int ox = (int)(id % OW4) * 4, oy = (int)((id / OW4) % OH);
  const int64_t b = id / ((int64_t)OW4 * OH);
  float acc[4][32];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int co = 0; co < 32; ++co) acc[p][co] = 0.f;
  const int ix0 = ox * 2 - 2;
  // the 18 input pairs of row ky + 1 are requested in front of row ky's 1 152 packed FMAs (requested inside the loop body they were
  // waited for where they were issued: five exposed round trips per row with two waves per SIMD to cover them)
  f32x2 raw[3][6];
  int okm[6], rowok_n;
  auto request = [&](int ky) __attribute__((always_inline)) {
    const int iy = oy * 2 - 2 + ky;
    rowok_n = (int)(iy >= 0) & (int)(iy < H);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int ix = ix0 + 2 * j;
      okm[j] = rowok_n & (int)(ix >= 0) & (int)(ix < W);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* xr = x + ((b * 3 + c) * H + (rowok_n ? iy : 0)) * (int64_t)W;
#pragma unroll
      for (int j = 0; j < 6; ++j) raw[c][j] = *reinterpret_cast<const f32x2*>(xr + (okm[j] ? ix0 + 2 * j : 0));
    }
  };
  request(0);
#pragma unroll 1
  for (int ky = 0; ky < 6; ++ky) {
    float v[3][12];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < 6; ++j) { v[c][2 * j] = okm[j] ? raw[c][j][0] : 0.f; v[c][2 * j + 1] = okm[j] ? raw[c][j][1] : 0.f; }
    request(ky < 5 ? ky + 1 : 5);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kx = 0; kx < 6; ++kx)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* wr = wt + ((ky * 6 + kx) * 3 + c) * 32;     // uniform address: scalar loads
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + 4 * q);
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[p][4 * q + e] = fmaf(v[c][kx + 2 * p], wv[e], acc[p][4 * q + e]);
        }
      }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (ox + p >= OW) break;
    float* o = out + ((b * OH + oy) * (int64_t)OW + ox + p) * out_ld + out_off;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * q);
      f32x4 r = {acc[p][4 * q] + bv[0], acc[p][4 * q + 1] + bv[1], acc[p][4 * q + 2] + bv[2], acc[p][4 * q + 3] + bv[3]};
      if (silu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = silu_fast(r[e]);
      }
      *reinterpret_cast<f32x4*>(o + 4 * q) = r;
    }
  }
}

__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ in, int in_ld, int in_off, float* __restrict__ out, int out_ld,
                                                         int out_off, int B, int H, int W, int C) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int C4 = C / 4;
  const int64_t total = (int64_t)B * 2 * H * 2 * W * C4;
  if (id >= total) return;
  const int c4 = (int)(id % C4);
  const int64_t p = id / C4;
  const int ox = (int)(p % (2 * W)), oy = (int)((p / (2 * W)) % (2 * H));
  const int64_t b = p / ((int64_t)4 * W * H);
  const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((b * H + (oy >> 1)) * W + (ox >> 1)) * in_ld + in_off + c4 * 4);
  *reinterpret_cast<f32x4*>(out + p * out_ld + out_off + c4 * 4) = v;
}

__global__ __launch_bounds__(256) void maxpool5_kernel(const float* __restrict__ in, int in_ld, int in_off, float* __restrict__ out, int out_ld,
                                                       int out_off, int B, int H, int W, int C) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int C4 = C / 4;
  const int64_t total = (int64_t)B * H * W * C4;
  if (id >= total) return;
  const int c4 = (int)(id % C4);
  const int64_t p = id / C4;
  const int ox = (int)(p % W), oy = (int)((p / W) % H);
  const int64_t b = p / ((int64_t)W * H);
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  for (int ky = -2; ky <= 2; ++ky)
    for (int kx = -2; kx <= 2; ++kx) {
      const int iy = oy + ky, ix = ox + kx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((b * H + iy) * W + ix) * in_ld + in_off + c4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    }
  *reinterpret_cast<f32x4*>(out + p * out_ld + out_off + c4 * 4) = m;
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

// one thread per (image, anchor, y, x, o)
__global__ __launch_bounds__(256) void yolo_decode_kernel(const float* __restrict__ raw, int raw_ld, float* __restrict__ pred, int B, int ny, int nx,
                                                          int na, int no, float stride, float aw0, float ah0, float aw1, float ah1, float aw2,
                                                          float ah2, int64_t total, int64_t row0) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t cnt = (int64_t)B * na * ny * nx * no;
  if (id >= cnt) return;
  const int o = (int)(id % no);
  int64_t t = id / no;
  const int x = (int)(t % nx); t /= nx;
  const int y = (int)(t % ny); t /= ny;
  const int an = (int)(t % na);
  const int64_t b = t / na;
  const float y0 = sigmoidf(raw[((b * ny + y) * nx + x) * raw_ld + an * no + o]);
  float v = y0;
  if (o == 0) v = (y0 * 2.0f + ((float)x - 0.5f)) * stride;          // (y * 2 + grid) * stride, grid = index - 0.5 (Detect._make_grid)
  else if (o == 1) v = (y0 * 2.0f + ((float)y - 0.5f)) * stride;
  else if (o == 2) { const float aw = an == 0 ? aw0 : (an == 1 ? aw1 : aw2); const float s2 = y0 * 2.0f; v = s2 * s2 * aw; }
  else if (o == 3) { const float ah = an == 0 ? ah0 : (an == 1 ? ah1 : ah2); const float s2 = y0 * 2.0f; v = s2 * s2 * ah; }
  pred[(b * total + row0 + ((int64_t)an * ny + y) * nx + x) * no + o] = v;
}

// cv2.resize(src, (new_w, new_h), INTER_LINEAR) on uint8 (OpenCV resize.cpp, restated): source coordinate
// fx = (dx + 0.5) * (src_w / new_w) - 0.5, sx = floor(fx), clamped; weights rounded to 11-bit fixed point
// (cvRound((1 - f) * 2048), cvRound(f * 2048)); horizontal pass in int (x 2^11), vertical pass
// ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  Then the 114 border, channel order, /255, CHW.
__device__ __forceinline__ void lin_coef(int d, int src, int dst, int& s0, int& s1, int& a0, int& a1) {
  const double scale = (double)src / (double)dst;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= src - 1) { f = 0.f; s = src - 1; }
  s0 = s; s1 = s + 1 < src ? s + 1 : src - 1;
  a0 = (int)rintf((1.f - f) * 2048.f);
  a1 = (int)rintf(f * 2048.f);
}

__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ img, int H, int W, int64_t rs, int bgr, int out_h, int out_w,
                                                        int new_h, int new_w, int top, int left, float fill, float* __restrict__ out) {
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= out_h * out_w) return;
  const int ox = id % out_w, oy = id / out_w;
  const int dx = ox - left, dy = oy - top;
  float v[3] = {fill, fill, fill};
  if (dx >= 0 && dx < new_w && dy >= 0 && dy < new_h) {
    if (new_w == W && new_h == H) {                       // letterbox skips the resize when the size already matches
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = (float)img[(int64_t)dy * rs + dx * 3 + c];
    } else {
      int x0, x1, ax0, ax1, y0, y1, by0, by1;
      lin_coef(dx, W, new_w, x0, x1, ax0, ax1);
      lin_coef(dy, H, new_h, y0, y1, by0, by1);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int S0 = (int)img[(int64_t)y0 * rs + x0 * 3 + c] * ax0 + (int)img[(int64_t)y0 * rs + x1 * 3 + c] * ax1;
        const int S1 = (int)img[(int64_t)y1 * rs + x0 * 3 + c] * ax0 + (int)img[(int64_t)y1 * rs + x1 * 3 + c] * ax1;
        const int r = (((by0 * (S0 >> 4)) >> 16) + ((by1 * (S1 >> 4)) >> 16) + 2) >> 2;
        v[c] = (float)(r < 0 ? 0 : (r > 255 ? 255 : r));
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) out[((int64_t)(bgr ? 2 - c : c) * out_h + oy) * out_w + ox] = v[c] / 255.0f;
}

// ---------------------------------------------------------------------------------------------------------------- NMS
// Workspace: [counter int x 64 | cand (row index, conf, cls) | sorted rows (offset box 4, conf, cls, raw box 4) | mask words]
struct NmsWs { size_t counter, cand_i, cand_s, cand_c, srt, mask, total; };
NmsWs nms_ws(int n, int max_nms) {
  NmsWs w; size_t off = 0;
  auto take = [&](size_t b) { const size_t o = off; off = align_up(off + b, 256); return o; };
  const size_t m = (size_t)(n < max_nms ? n : max_nms);
  w.counter = take(256);
  w.cand_i = take((size_t)n * 4); w.cand_s = take((size_t)n * 4); w.cand_c = take((size_t)n * 4);
  w.srt = take(m * 10 * 4);
  w.mask = take(m * ((m + 63) / 64) * 8);
  w.total = off;
  return w;
}

// candidates: objectness > thr, then conf = obj * max class prob > thr (non-multi-label branch: best class only)
__global__ __launch_bounds__(256) void nms_filter_kernel(const float* __restrict__ pred, int n, int nc, float thr, int* __restrict__ counter,
                                                         int* __restrict__ ci, float* __restrict__ cs, int* __restrict__ cc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = pred + (int64_t)i * (5 + nc);
  const float obj = p[4];
  if (!(obj > thr)) return;
  float best = -INFINITY; int bj = 0;
  for (int j = 0; j < nc; ++j) { const float c = p[5 + j] * obj; if (c > best) { best = c; bj = j; } }   // first maximum wins
  if (!(best > thr)) return;
  const int pos = atomicAdd(counter, 1);
  ci[pos] = i; cs[pos] = best; cc[pos] = bj;
}

// rank by (confidence descending, row index ascending) — counting sort, O(m^2) compares spread over m threads;
// rows ranked >= max_nms are dropped (the reference keeps the 30000 most confident)
__global__ __launch_bounds__(256) void nms_rank_kernel(const float* __restrict__ pred, int nc, const int* __restrict__ counter, const int* __restrict__ ci,
                                                       const float* __restrict__ cs, const int* __restrict__ cc, int max_nms, float max_wh,
                                                       int agnostic, float* __restrict__ srt) {
  const int m = *counter;
  if ((int)blockIdx.x * 256 >= m) return;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const bool live = p < m;
  const float s = live ? cs[p] : 0.f; const int i = live ? ci[p] : 0;
  // every thread compares against every candidate: the candidates go through LDS in tiles (one broadcast read per compare
  // instead of two global loads)
  __shared__ float ts[1024];
  __shared__ int ti[1024];
  int rank = 0;
  for (int q0 = 0; q0 < m; q0 += 1024) {
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += 256) {
      const int q = q0 + t;
      ts[t] = q < m ? cs[q] : -INFINITY;                  // (-inf, INT_MAX) never outranks anything
      ti[t] = q < m ? ci[q] : 0x7fffffff;
    }
    __syncthreads();
#pragma unroll 8
    for (int t = 0; t < 1024; ++t) {
      const float sq = ts[t];
      rank += (sq > s || (sq == s && ti[t] < i)) ? 1 : 0;
    }
  }
  if (!live || rank >= max_nms) return;
  const float* b = pred + (int64_t)i * (5 + nc);
  const float x1 = b[0] - b[2] / 2, y1 = b[1] - b[3] / 2, x2 = b[0] + b[2] / 2, y2 = b[1] + b[3] / 2;   // xywh2xyxy
  const float off = agnostic ? 0.f : (float)cc[p] * max_wh;
  float* o = srt + (int64_t)rank * 10;
  o[0] = x1 + off; o[1] = y1 + off; o[2] = x2 + off; o[3] = y2 + off; o[4] = s; o[5] = (float)cc[p];
  o[6] = x1; o[7] = y1; o[8] = x2; o[9] = y2;
}

// mask[i][w] bit j: box 64w + j (ranked after i) overlaps box i with IoU > thr (torchvision.ops.nms: inter / (a_i + a_j - inter))
__global__ __launch_bounds__(256) void nms_mask_kernel(const float* __restrict__ srt, const int* __restrict__ counter, int max_nms, float thr,
                                                       unsigned long long* __restrict__ mask) {
  int m = *counter; m = m < max_nms ? m : max_nms;
  const int nw = (m + 63) / 64;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (int64_t)m * nw) return;
  const int i = (int)(id / nw), w = (int)(id % nw);
  const float* a = srt + (int64_t)i * 10;
  const float ax1 = a[0], ay1 = a[1], ax2 = a[2], ay2 = a[3];
  const float aa = (ax2 - ax1) * (ay2 - ay1);
  unsigned long long bits = 0ull;
  for (int j = 0; j < 64; ++j) {
    const int q = w * 64 + j;
    if (q <= i || q >= m) continue;
    const float* b = srt + (int64_t)q * 10;
    const float iw = fminf(ax2, b[2]) - fmaxf(ax1, b[0]), ih = fminf(ay2, b[3]) - fmaxf(ay1, b[1]);
    const float inter = fmaxf(iw, 0.f) * fmaxf(ih, 0.f);
    const float ab = (b[2] - b[0]) * (b[3] - b[1]);
    if (inter / (aa + ab - inter) > thr) bits |= 1ull << j;
  }
  mask[id] = bits;
}

// greedy scan in rank order by ONE wave: lane l holds the words l, l+64, ... of the `removed` bit set.  Only the mask rows of
// boxes that are still unsuppressed are ever loaded (a localizer that fires on every anchor ranks 25 200 boxes and keeps ~70:
// reading all 25 200 rows of 3 KB cost 1.4 ms per image): the next NMS_G unsuppressed boxes under the current bit set are found
// first, their rows requested together (one memory latency per group), then they are taken in order, each re-checked against
// the rows OR-ed in by its predecessors in the group.
constexpr int NMS_WPL = 8;                                // words per lane: up to 64 * 8 * 64 = 32768 ranked boxes
constexpr int NMS_G = 4;
__global__ __launch_bounds__(64) void nms_scan_kernel(const float* __restrict__ srt, const int* __restrict__ counter, int max_nms, int max_det,
                                                      const unsigned long long* __restrict__ mask, float* __restrict__ out, int* __restrict__ count) {
  int m = *counter; m = m < max_nms ? m : max_nms;
  const int nw = (m + 63) / 64;
  const int lane = threadIdx.x;
  unsigned long long rem[NMS_WPL];
#pragma unroll
  for (int t = 0; t < NMS_WPL; ++t) rem[t] = 0ull;
  auto word_of = [&](int w) -> unsigned long long {      // word w of the bit set, to every lane
    const int t = w >> 6, src = w & 63;
    unsigned long long word = 0ull;
#pragma unroll
    for (int tt = 0; tt < NMS_WPL; ++tt) if (tt == t) word = rem[tt];
    const unsigned lo = __shfl((unsigned)(word & 0xffffffffull), src, 64), hi = __shfl((unsigned)(word >> 32), src, 64);
    return ((unsigned long long)hi << 32) | lo;
  };
  int kept = 0, pos = 0;                                  // every box below pos is decided
  while (pos < m && kept < max_det) {
    // the next (up to) NMS_G boxes >= pos that are unsuppressed NOW
    int cand[NMS_G], nc = 0, w = pos >> 6;
    unsigned long long free_bits = ~word_of(w) & (~0ull << (pos & 63));
#pragma unroll
    for (int r = 0; r < NMS_G; ++r) {                     // (unrolled: cand[] stays in registers)
      cand[r] = -1;
      while (w < nw) {
        if (w == nw - 1 && (m & 63)) free_bits &= (1ull << (m & 63)) - 1ull;   // bits past the last box
        if (free_bits != 0ull) break;
        if (++w < nw) free_bits = ~word_of(w);
      }
      if (w < nw) {
        cand[r] = w * 64 + __builtin_ctzll(free_bits);
        free_bits &= free_bits - 1ull;
        nc = r + 1;
      }
    }
    if (nc == 0) break;
    unsigned long long row[NMS_G][NMS_WPL];
#pragma unroll
    for (int r = 0; r < NMS_G; ++r)
#pragma unroll
      for (int t = 0; t < NMS_WPL; ++t) {
        const int ww = lane + 64 * t;
        row[r][t] = (cand[r] >= 0 && ww < nw) ? mask[(int64_t)cand[r] * nw + ww] : 0ull;
      }
#pragma unroll
    for (int r = 0; r < NMS_G; ++r) {
      const int i = cand[r];
      if (i < 0 || kept >= max_det) break;
      if ((word_of(i >> 6) >> (i & 63)) & 1ull) continue;  // suppressed by a predecessor of this group
      if (lane < 6) {
        const float* sr = srt + (int64_t)i * 10;
        out[(int64_t)kept * 6 + lane] = lane < 4 ? sr[6 + lane] : sr[lane];
      }
      ++kept;
#pragma unroll
      for (int tt = 0; tt < NMS_WPL; ++tt) rem[tt] |= row[r][tt];
    }
    int last = cand[0];
#pragma unroll
    for (int r = 1; r < NMS_G; ++r) last = cand[r] >= 0 ? cand[r] : last;
    pos = last + 1;
  }
  if (lane == 0) *count = kept;
}


// ---- few kept boxes out of many candidates (text lines: max_det of tens, a detector that fires on thousands of anchors): greedy NMS
// WITHOUT the sort and WITHOUT the m x m mask.  torchvision's result is "walk the boxes by descending score, keep a box iff no kept
// box overlaps it with IoU > thr"; the same set comes out of: keep the best live candidate, kill every live candidate it overlaps,
// repeat — each round one pass over the candidates, which never leave the registers (boxes) / LDS (scores) of ONE workgroup per image:
//   work = kept x candidates IoUs (64 x 25 200) instead of candidates^2 / 2 compares + candidates^2 / 2 IoUs (6.4e8 each),
// and the images of a batch run side by side (one launch, grid = images).  Same arithmetic as the three-kernel path: the filter of
// nms_filter_kernel, the boxes of nms_rank_kernel, the IoU expression of nms_mask_kernel, ties by ascending row index.
constexpr int NG_T = 512, NG_PER = 50;                    // 2 waves per SIMD (256 registers): 4 x 50 box registers per thread; n <= 25 600
__global__ __launch_bounds__(NG_T) void nms_greedy_kernel(const float* __restrict__ pred_all, int n, int nc, float conf_thr, float iou_thr, int max_det,
                                                          float max_wh, int agnostic, float* __restrict__ out_all, int* __restrict__ count_all) {
  const float* pred = pred_all + (int64_t)blockIdx.x * n * (5 + nc);
  float* out = out_all + (int64_t)blockIdx.x * max_det * 6;
  __shared__ float sc[NG_PER * NG_T];                     // score of candidate (slot j, thread t) = row j * NG_T + t; -inf: dead
  __shared__ float red_s[NG_T / 64];
  __shared__ int red_r[NG_T / 64];
  __shared__ float kb[4];                                 // the box kept in this round (offset form)
  __shared__ int kr;                                      // its row (-1: nothing left)
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  float bx[NG_PER][4];
#pragma unroll
  for (int j = 0; j < NG_PER; ++j) {
    const int r = j * NG_T + t;
    float s = -INFINITY;
    bx[j][0] = bx[j][1] = bx[j][2] = bx[j][3] = 0.f;
    if (r < n) {
      const float* p = pred + (int64_t)r * (5 + nc);
      const float obj = p[4];
      if (obj > conf_thr) {
        float best = -INFINITY; int bj = 0;
        for (int c = 0; c < nc; ++c) { const float v = p[5 + c] * obj; if (v > best) { best = v; bj = c; } }   // first maximum wins
        if (best > conf_thr) {
          s = best;
          const float x1 = p[0] - p[2] / 2, y1 = p[1] - p[3] / 2, x2 = p[0] + p[2] / 2, y2 = p[1] + p[3] / 2;   // xywh2xyxy
          const float off = agnostic ? 0.f : (float)bj * max_wh;
          bx[j][0] = x1 + off; bx[j][1] = y1 + off; bx[j][2] = x2 + off; bx[j][3] = y2 + off;
        }
      }
    }
    sc[j * NG_T + t] = s;
  }
  int kept = 0;
  float ax1 = 0.f, ay1 = 0.f, ax2 = 0.f, ay2 = 0.f, aa = 0.f;
  bool have = false;                                      // a box was kept in the previous round: its kills are applied in this pass
  while (true) {
    // one pass: apply the previous winner's kills, find this thread's best live candidate
    float bs = -INFINITY; int br = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NG_PER; ++j) {
      float s = sc[j * NG_T + t];
      if (s > -INFINITY) {
        if (have) {
          const float iw = fminf(ax2, bx[j][2]) - fmaxf(ax1, bx[j][0]), ih = fminf(ay2, bx[j][3]) - fmaxf(ay1, bx[j][1]);
          const float inter = fmaxf(iw, 0.f) * fmaxf(ih, 0.f);
          const float ab = (bx[j][2] - bx[j][0]) * (bx[j][3] - bx[j][1]);
          if (inter / (aa + ab - inter) > iou_thr) { s = -INFINITY; sc[j * NG_T + t] = s; }
        }
        if (s > bs) { bs = s; br = j * NG_T + t; }       // (rows ascend with j: the first of equal scores stays)
      }
    }
    if (kept >= max_det) break;                           // (uniform)
    // workgroup argmax by (score descending, row ascending)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const float os = __shfl_xor(bs, d, 64); const int orw = __shfl_xor(br, d, 64);
      if (os > bs || (os == bs && orw < br)) { bs = os; br = orw; }
    }
    if (lane == 0) { red_s[wv] = bs; red_r[wv] = br; }
    __syncthreads();
    if (wv == 0) {
      float s2 = lane < NG_T / 64 ? red_s[lane] : -INFINITY; int r2 = lane < NG_T / 64 ? red_r[lane] : 0x7fffffff;
#pragma unroll
      for (int d = 4; d >= 1; d >>= 1) {
        const float os = __shfl_xor(s2, d, 64); const int orw = __shfl_xor(r2, d, 64);
        if (os > s2 || (os == s2 && orw < r2)) { s2 = os; r2 = orw; }
      }
      if (lane == 0) kr = s2 > -INFINITY ? r2 : -1;
    }
    __syncthreads();
    const int win = kr;
    if (win < 0) break;                                   // (uniform)
    if ((win & (NG_T - 1)) == t) {                        // the owner publishes the box and retires the candidate
      const int jw = win / NG_T;
#pragma unroll
      for (int j = 0; j < NG_PER; ++j)
        if (j == jw) { kb[0] = bx[j][0]; kb[1] = bx[j][1]; kb[2] = bx[j][2]; kb[3] = bx[j][3]; sc[j * NG_T + t] = -INFINITY; }
      const float* p = pred + (int64_t)win * (5 + nc);
      const float obj = p[4];
      float best = -INFINITY; int bj = 0;
      for (int c = 0; c < nc; ++c) { const float v = p[5 + c] * obj; if (v > best) { best = v; bj = c; } }
      float* o = out + (int64_t)kept * 6;
      o[0] = p[0] - p[2] / 2; o[1] = p[1] - p[3] / 2; o[2] = p[0] + p[2] / 2; o[3] = p[1] + p[3] / 2; o[4] = best; o[5] = (float)bj;
    }
    __syncthreads();
    ax1 = kb[0]; ay1 = kb[1]; ax2 = kb[2]; ay2 = kb[3];
    aa = (ax2 - ax1) * (ay2 - ay1);
    have = true;
    ++kept;
  }
  if (t == 0) count_all[blockIdx.x] = kept;
}

}  // namespace

int im2col_nchw(const float* x, float* col, int B, int Cin, int H, int W, int KH, int KW, int stride, int pad, int OH, int OW, int kpad, hipStream_t s) {
  const int64_t total = (int64_t)B * OH * OW * kpad;
  if (total <= 0) return EFFOCR_OK;
  if (kpad < KH * KW * Cin) return fail(EFFOCR_EINVAL, "im2col: padded K smaller than the tap count");
  hipLaunchKernelGGL(im2col_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, col, B, Cin, H, W, KH, KW, stride, pad, OH, OW, kpad);
  return check_launch("im2col_nchw");
}

int stem6x6s2_nchw(const float* x, const float* w, int w_ld, const float* wt, const float* bias, float* out, int B, int H, int W, int OH, int OW, int out_ld,
                   int out_off, int silu, hipStream_t s) {
  const int64_t total = (int64_t)B * OH * ((OW + 1) / 2);
  if (total <= 0) return EFFOCR_OK;
  if (w_ld < 108 || ((out_ld | out_off) & 3)) return fail(EFFOCR_EINVAL, "stem conv: weight rows of >= 108 taps, channel stride / offset multiples of 4");
  if (wt && (W & 1) == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0) {      // transposed weights + 8-byte input pairs: the scalar-weight kernel
    const int64_t t4 = (int64_t)B * OH * ((OW + 3) / 4);
    hipLaunchKernelGGL(stem6x6s2_sw_kernel, dim3((unsigned)((t4 + 255) / 256)), dim3(256), 0, s, x, wt, bias, out, B, H, W, OH, OW, out_ld, out_off, silu);
    return check_launch("stem6x6s2_sw");
  }
  hipLaunchKernelGGL(stem6x6s2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, w_ld, bias, out, B, H, W, OH, OW, out_ld, out_off, silu);
  return check_launch("stem6x6s2");
}

int upsample2x_nhwc(const float* in, int in_ld, int in_off, float* out, int out_ld, int out_off, int B, int H, int W, int C, hipStream_t s) {
  const int64_t total = (int64_t)B * 4 * H * W * (C / 4);
  if (total <= 0) return EFFOCR_OK;
  if ((C | in_ld | in_off | out_ld | out_off) & 3) return fail(EFFOCR_EUNSUPPORTED, "upsample: channel counts / strides / offsets must be multiples of 4");
  hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, in_ld, in_off, out, out_ld, out_off, B, H, W, C);
  return check_launch("upsample2x");
}

int maxpool5_nhwc(const float* in, int in_ld, int in_off, float* out, int out_ld, int out_off, int B, int H, int W, int C, hipStream_t s) {
  const int64_t total = (int64_t)B * H * W * (C / 4);
  if (total <= 0) return EFFOCR_OK;
  if ((C | in_ld | in_off | out_ld | out_off) & 3) return fail(EFFOCR_EUNSUPPORTED, "maxpool5: channel counts / strides / offsets must be multiples of 4");
  hipLaunchKernelGGL(maxpool5_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, in_ld, in_off, out, out_ld, out_off, B, H, W, C);
  return check_launch("maxpool5");
}

int yolo_decode(const float* raw, int raw_ld, float* pred, int B, int ny, int nx, int na, int no, float stride, const float* anchors_px,
                int64_t total, int64_t row0, hipStream_t s) {
  const int64_t cnt = (int64_t)B * na * ny * nx * no;
  if (cnt <= 0) return EFFOCR_OK;
  if (na != 3) return fail(EFFOCR_EUNSUPPORTED, "yolo_decode: three anchors per level");
  hipLaunchKernelGGL(yolo_decode_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, raw, raw_ld, pred, B, ny, nx, na, no, stride,
                     anchors_px[0], anchors_px[1], anchors_px[2], anchors_px[3], anchors_px[4], anchors_px[5], total, row0);
  return check_launch("yolo_decode");
}

int letterbox_u8(const uint8_t* img, int H, int W, int64_t row_stride, int bgr, int out_h, int out_w, int new_h, int new_w, int top, int left,
                 float fill, float* out, hipStream_t s) {
  if (H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0 || new_h <= 0 || new_w <= 0 || top < 0 || left < 0 || top + new_h > out_h || left + new_w > out_w)
    return fail(EFFOCR_EINVAL, "letterbox: bad geometry");
  hipLaunchKernelGGL(letterbox_kernel, dim3((unsigned)((out_h * out_w + 255) / 256)), dim3(256), 0, s, img, H, W, row_stride, bgr, out_h, out_w,
                     new_h, new_w, top, left, fill, out);
  return check_launch("letterbox");
}

size_t nms_workspace_bytes(int n, int max_nms) { return n <= 0 ? 256 : nms_ws(n, max_nms).total; }

int nms_yolo(const float* pred, int n, int nc, float conf_thres, float iou_thres, int max_det, int max_nms, float max_wh, int agnostic,
             float* out, int* count, void* ws, size_t ws_bytes, hipStream_t s) {
  if (n < 0 || nc < 1 || max_det < 1 || max_nms < 1) return fail(EFFOCR_EINVAL, "nms: bad sizes");
  if (!(conf_thres >= 0.f && conf_thres <= 1.f) || !(iou_thres >= 0.f && iou_thres <= 1.f)) return fail(EFFOCR_EINVAL, "nms: thresholds must lie in [0, 1]");
  if (max_nms > 64 * 64 * NMS_WPL) return fail(EFFOCR_EUNSUPPORTED, "nms: at most 32768 ranked boxes");
  if (n == 0) return hipMemsetAsync(count, 0, 4, s) == hipSuccess ? EFFOCR_OK : fail(EFFOCR_EHIP, "nms: memset failed");
  const NmsWs w = nms_ws(n, max_nms);
  if (ws_bytes < w.total) return fail(EFFOCR_EWORKSPACE, "nms: workspace too small");
  char* W = static_cast<char*>(ws);
  int* counter = reinterpret_cast<int*>(W + w.counter);
  int* ci = reinterpret_cast<int*>(W + w.cand_i); float* cs = reinterpret_cast<float*>(W + w.cand_s); int* cc = reinterpret_cast<int*>(W + w.cand_c);
  float* srt = reinterpret_cast<float*>(W + w.srt);
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(W + w.mask);
  if (hipMemsetAsync(counter, 0, 256, s) != hipSuccess) return fail(EFFOCR_EHIP, "nms: memset failed");
  const unsigned gb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(nms_filter_kernel, dim3(gb), dim3(256), 0, s, pred, n, nc, conf_thres, counter, ci, cs, cc);
  hipLaunchKernelGGL(nms_rank_kernel, dim3(gb), dim3(256), 0, s, pred, nc, counter, ci, cs, cc, max_nms, max_wh, agnostic, srt);
  const int mcap = n < max_nms ? n : max_nms;
  const int64_t words = (int64_t)mcap * ((mcap + 63) / 64);
  hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, srt, counter, max_nms, iou_thres, mask);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 0, s, srt, counter, max_nms, max_det, mask, out, count);
  return check_launch("nms");
}

// B images, pred [B][n][5 + nc] -> out [B][max_det][6], count [B].  One launch of the greedy kernel when it applies (every candidate of an
// image fits one workgroup and few boxes are kept), else the three-kernel path image by image (one workspace, stream order).
// (Round 4: no max_det bound any more — the reference's own max_det is 1000 (localizer_engine.py:62) and the product default must take this
// path.  Cost is kept x one pass (~2-3 us): a text line keeps tens of boxes; a pathological image that keeps all 1000 costs ~3 ms for the
// whole batch, images side by side, where the per-image path takes 0.24 ms per image one after the other.)
bool nms_greedy_applies(int n, int max_det, int max_nms) { (void)max_det; return n <= NG_T * NG_PER && n <= max_nms; }

int nms_yolo_batch(const float* pred, int B, int n, int nc, float conf_thres, float iou_thres, int max_det, int max_nms, float max_wh, int agnostic,
                   float* out, int* count, void* ws, size_t ws_bytes, hipStream_t s) {
  if (B < 0 || n < 0 || nc < 1 || max_det < 1 || max_nms < 1) return fail(EFFOCR_EINVAL, "nms: bad sizes");
  if (!(conf_thres >= 0.f && conf_thres <= 1.f) || !(iou_thres >= 0.f && iou_thres <= 1.f)) return fail(EFFOCR_EINVAL, "nms: thresholds must lie in [0, 1]");
  if (B == 0) return EFFOCR_OK;
  if (n > 0 && nms_greedy_applies(n, max_det, max_nms)) {
    hipLaunchKernelGGL(nms_greedy_kernel, dim3((unsigned)B), dim3(NG_T), 0, s, pred, n, nc, conf_thres, iou_thres, max_det, max_wh, agnostic, out, count);
    return check_launch("nms_greedy");
  }
  for (int b = 0; b < B; ++b) {
    const int rc = nms_yolo(pred + (int64_t)b * n * (5 + nc), n, nc, conf_thres, iou_thres, max_det, max_nms, max_wh, agnostic, out + (int64_t)b * max_det * 6,
                            count + b, ws, ws_bytes, s);
    if (rc) return rc;
  }
  return EFFOCR_OK;
}

}  // namespace effocr
