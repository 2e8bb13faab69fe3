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

struct RowCoord { int b, iy0, ix0; };

__device__ __forceinline__ void conv_stage_load(u32x4 (&r)[4], const ConvArgs& a, const RowCoord (&rc)[4], int ks, int tid) {
  const int c = tid & 7;
  const int kk = ks * 32;
  const int tap = kk / a.Cin, ci0 = kk - tap * a.Cin;
  const int ky = tap / a.KW, kx = tap - ky * a.KW;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int iy = rc[i].iy0 + ky, ix = rc[i].ix0 + kx;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
      const float* p = a.in + (((int64_t)rc[i].b * a.H + iy) * a.W + ix) * a.in_ld + a.in_off + ci0 + c * 4;
      v = *reinterpret_cast<const u32x4*>(p);
    }
    r[i] = v;
  }
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

// E = float: fp32 operands on v_mfma_f32_32x32x2_f32 (exact products).  E = __bf16: operands rounded to bf16 (activations in the
// stage loader, weights once at upload: a.w16 [Cout][ceil(K/64)*64]) on v_mfma_f32_32x32x16_bf16, fp32 accumulation and epilogue.
template <typename E>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
  constexpr bool B16 = !tile128::is_f32<E>::value;
  __shared__ __attribute__((aligned(16))) char smem[GEMM_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = wave_id(), wn = w >> 1, wm = w & 1;
  const int M = a.B * a.OH * a.OW;
  const int K = a.KH * a.KW * a.Cin;
  const int ntn = (a.Cout + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;
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
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 rw[4], rx[4];
  auto load_stage = [&](int ks) __attribute__((always_inline)) {
    if constexpr (B16) {
      stage_load<__bf16>(rw, static_cast<const __bf16*>(a.w16), Kp, n0, a.Cout, ks * ROWB, tid);
      conv_stage_load16(rx, a, rc, ks, K, tid);
    } else {
      stage_load<float>(rw, a.w, K, n0, a.Cout, ks * ROWB, tid);
      conv_stage_load(rx, a, rc, ks, tid);
    }
  };
  load_stage(ks_lo);
  stage_store<E>(rw, smem, tid);
  stage_store<E>(rx, smem + TILEB, tid);
  __syncthreads();
  for (int ks = ks_lo; ks < ks_hi; ++ks) {
    char* cur = smem + ((ks - ks_lo) & 1) * STAGEB;
    char* nxt = smem + (((ks - ks_lo) & 1) ^ 1) * STAGEB;
    const bool more = (ks + 1) < ks_hi;
    if (more) load_stage(ks + 1);
    stage_mma<E>(acc, cur, cur + TILEB, wn, wm, lane);
    if (more) {
      stage_store<E>(rw, nxt, tid);
      stage_store<E>(rx, nxt + TILEB, tid);
    }
    __syncthreads();
  }

  const int half = lane >> 5;
  if (ksp > 1) {                                         // raw partial sums [split][M][Cout]; conv_reduce_kernel finishes
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + wm * 64 + j * 32 + (lane & 31);
      if (m >= M) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 64 + i * 32 + 8 * q + 4 * half;
          if (n >= a.Cout) continue;
          const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
          *reinterpret_cast<f32x4*>(a.partial + ((int64_t)sp * M + m) * a.Cout + n) = v;
        }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wm * 64 + j * 32 + (lane & 31);
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + i * 32 + 8 * q + 4 * half;
        if (n >= a.Cout) continue;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + n);
        f32x4 v = {acc[i][j][4 * q] + bv[0], acc[i][j][4 * q + 1] + bv[1], acc[i][j][4 * q + 2] + bv[2], acc[i][j][4 * q + 3] + bv[3]};
        if (a.silu) {                                    // x * sigmoid(x), before the residual (Bottleneck: x + cv2(cv1(x)))
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
        }
        if (a.resid) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(a.resid + (int64_t)m * a.res_ld + a.res_off + n);
          v += rv;
        }
        if (a.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(a.out + (int64_t)m * a.out_ld + a.out_off + n) = v;
      }
    }
  }
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
    for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
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
  const int64_t grid = ((M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
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
  if (a.w16) hipLaunchKernelGGL(conv_igemm_kernel<__bf16>, g, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(conv_igemm_kernel<float>, g, dim3(256), 0, s, a);
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
