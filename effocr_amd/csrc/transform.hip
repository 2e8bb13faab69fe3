// Crop pre-processing on the device: the reference's per-character `create_paired_transform`
// (utils/datasets_utils.py:69-90,166-172, called per box at infer_effocr.py:286-293) as ONE kernel over a
// whole box list:   uint8 HWC page/line image + boxes  ->  [n, 3, S, S] fp32, ImageNet-normalised.
//
//   crop im[y0:y1, x0:x1]  ->  pad right/bottom to a square of side L = max(h, w) with `fill`  ->  /255
//   ->  bilinear resize L x L -> S x S (align_corners = False, antialias on/off)  ->  (v - mean) / std
//
// Nothing is materialised: a thread owns 4 consecutive output pixels of one row (float4 stores, 3 planes)
// and evaluates the separable filter straight from the uint8 image (the crop is a few KB: L1/L2 hits),
// width first, then height, as ATen does (fp32 row intermediate).  The /255 is applied to the filtered
// value instead of every tap (weights sum to 1; <= 2 ulp from ATen's order).  HBM-write bound:
// 602 112 B per crop at S = 224 — 301 056 B with the 16-bit hand-off (SURVEY f-2: "-> [B,3,224,224] bf16"): the fp32 value rounded
// ONCE to the encoder's operand type, which is exactly what the patch embedding does with an fp32 crop before its MFMAs, so the
// tokens are bit-identical either way and half the bytes cross HBM twice.
#include "common.hpp"
#include "kernels.hpp"

namespace effocr {
namespace {

struct Taps { int start, count; float support_c, inv, norm; float center; };

// ATen _compute_indices_min_size_weights_aa (triangle filter): window [start, start+count), weight of tap
// j = max(0, 1 - |(j + start - center + 0.5) * inv|) * norm.
__device__ __forceinline__ Taps aa_taps(int i, float scale, int in_size) {
  Taps t;
  const float support = scale >= 1.f ? scale : 1.f;
  t.inv = scale >= 1.f ? 1.f / scale : 1.f;
  t.center = scale * ((float)i + 0.5f);
  int lo = (int)(t.center - support + 0.5f);
  lo = lo > 0 ? lo : 0;
  int hi = (int)(t.center + support + 0.5f);
  hi = hi < in_size ? hi : in_size;
  t.start = lo;
  t.count = hi - lo > 0 ? hi - lo : 0;
  float tot = 0.f;
  for (int j = 0; j < t.count; ++j) {
    const float w = 1.f - fabsf(((float)(j + lo) - t.center + 0.5f) * t.inv);
    tot += w > 0.f ? w : 0.f;
  }
  t.norm = tot != 0.f ? 1.f / tot : 0.f;
  t.support_c = support;
  return t;
}
__device__ __forceinline__ float aa_weight(const Taps& t, int j) {
  const float w = 1.f - fabsf(((float)(j + t.start) - t.center + 0.5f) * t.inv);
  return (w > 0.f ? w : 0.f) * t.norm;
}

// plain bilinear (area_pixel_compute_source_index, align_corners = False): taps (i0, i1) weights (1-l, l)
__device__ __forceinline__ void lin_taps(int i, float scale, int in_size, int& i0, int& i1, float& l) {
  float src = scale * ((float)i + 0.5f) - 0.5f;
  src = src > 0.f ? src : 0.f;
  i0 = (int)src;
  i0 = i0 < in_size - 1 ? i0 : in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l = src - (float)i0;
  l = l < 0.f ? 0.f : (l > 1.f ? 1.f : l);
}

struct CropArgs {
  const uint8_t* img; int H, W; int64_t stride;
  int64_t img_stride; int n_img, box_ld;                 // batch form: box_ld = 5, column 4 = image index, images img_stride bytes apart
  const int* boxes; int n; int S;
  void* out;
  float mean[3], std[3], fill[3];
};

template <bool AA, typename TO>
__global__ __launch_bounds__(256) void crop_transform_kernel(CropArgs a) {
  const int b = blockIdx.y;
  const int S = a.S, S4 = S >> 2;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= S * S4) return;
  const int oy = p / S4, ox = (p - oy * S4) * 4;
  const int* bx = a.boxes + (size_t)a.box_ld * b;
  int x0 = bx[0], y0 = bx[1], x1 = bx[2], y1 = bx[3];
  const int im = a.box_ld > 4 ? bx[4] : 0;
  x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0;
  x1 = x1 > a.W ? a.W : x1; y1 = y1 > a.H ? a.H : y1;
  const bool im_ok = im >= 0 && im < a.n_img;
  const int w = im_ok ? x1 - x0 : 0, h = y1 - y0;
  TO* o = static_cast<TO*>(a.out) + (size_t)b * 3 * S * S + (size_t)oy * S + ox;
  auto put4 = [&](int c, const f32x4& r) {                // 4 consecutive pixels of plane c: 16 bytes fp32 / 8 bytes 16-bit (one rounding)
    if constexpr (sizeof(TO) == 4) *reinterpret_cast<f32x4*>(o + (size_t)c * S * S) = r;
    else *reinterpret_cast<u32x2*>(o + (size_t)c * S * S) = pack4<TO>(r[0], r[1], r[2], r[3]);
  };
  if (w <= 0 || h <= 0) {                                 // host rejects empty boxes; never read out of bounds
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < 3; ++c) put4(c, z);
    return;
  }
  const int L = w > h ? w : h;
  const float scale = (float)L / (float)S;
  const uint8_t* base = a.img + (size_t)im * a.img_stride + (size_t)y0 * a.stride + (size_t)x0 * 3;

  float acc[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[q][c] = 0.f;

  // one padded-square pixel (raw 0..255 scale)
  auto px = [&](int y, int x, float (&v)[3]) {
    if (y < h && x < w) {
      const uint8_t* s = base + (size_t)y * a.stride + x * 3;
      v[0] = (float)s[0]; v[1] = (float)s[1]; v[2] = (float)s[2];
    } else { v[0] = a.fill[0]; v[1] = a.fill[1]; v[2] = a.fill[2]; }
  };

  if constexpr (AA) {
    const Taps ty = aa_taps(oy, scale, L);
    Taps tx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) tx[q] = aa_taps(ox + q, scale, L);
    for (int jy = 0; jy < ty.count; ++jy) {
      const int y = ty.start + jy;
      const float wy = aa_weight(ty, jy);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float row[3] = {0.f, 0.f, 0.f};
        for (int jx = 0; jx < tx[q].count; ++jx) {
          float v[3];
          px(y, tx[q].start + jx, v);
          const float wx = aa_weight(tx[q], jx);
          row[0] = fmaf(wx, v[0], row[0]); row[1] = fmaf(wx, v[1], row[1]); row[2] = fmaf(wx, v[2], row[2]);
        }
        acc[q][0] = fmaf(wy, row[0], acc[q][0]); acc[q][1] = fmaf(wy, row[1], acc[q][1]); acc[q][2] = fmaf(wy, row[2], acc[q][2]);
      }
    }
  } else {
    int ya, yb; float ly;
    lin_taps(oy, scale, L, ya, yb, ly);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int xa, xb; float lx;
      lin_taps(ox + q, scale, L, xa, xb, lx);
      float v00[3], v01[3], v10[3], v11[3];
      px(ya, xa, v00); px(ya, xb, v01); px(yb, xa, v10); px(yb, xb, v11);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float r0 = (1.f - lx) * v00[c] + lx * v01[c];
        const float r1 = (1.f - lx) * v10[c] + lx * v11[c];
        acc[q][c] = (1.f - ly) * r0 + ly * r1;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = (acc[q][c] / 255.f - a.mean[c]) / a.std[c];
    put4(c, r);
  }
}

}  // namespace

// n_img images of one geometry, img_stride bytes apart; boxes [n, box_ld] int32 (box_ld 4: every box cuts image 0; 5: column 4 =
// image index; a box that names no image yields a zero crop).  Any n: launched in slices of 65535 boxes.
int crop_transform(const uint8_t* img, int n_img, int64_t img_stride, int H, int W, int64_t stride, const int* boxes, int box_ld, int64_t n, int S,
                   int antialias, const float* mean, const float* stdv, const float* fill, void* out, int out_prec, hipStream_t s) {
  if (n <= 0) return EFFOCR_OK;
  if (out_prec != PREC_FP32 && out_prec != PREC_BF16 && out_prec != PREC_FP16) return fail(EFFOCR_EINVAL, "crop_transform: unknown output type");
  const size_t osz = out_prec == PREC_FP32 ? 4 : 2;
  if (S <= 0 || (S & 3)) return fail(EFFOCR_EUNSUPPORTED, "crop_transform: output size must be a positive multiple of 4");
  if (box_ld != 4 && box_ld != 5) return fail(EFFOCR_EINVAL, "crop_transform: boxes must have 4 or 5 columns");
  for (int64_t lo = 0; lo < n; lo += 65535) {
    const int m = (int)(n - lo < 65535 ? n - lo : 65535);
    CropArgs a;
    a.img = img; a.H = H; a.W = W; a.stride = stride; a.img_stride = img_stride; a.n_img = n_img; a.box_ld = box_ld;
    a.boxes = boxes + lo * box_ld; a.n = m; a.S = S; a.out = static_cast<char*>(out) + (size_t)lo * 3 * S * S * osz;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.std[c] = stdv[c]; a.fill[c] = fill[c]; }
    const dim3 grid((unsigned)((S * (S / 4) + 255) / 256), (unsigned)m);
    if (out_prec == PREC_FP32) {
      if (antialias) hipLaunchKernelGGL((crop_transform_kernel<true, float>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((crop_transform_kernel<false, float>), grid, dim3(256), 0, s, a);
    } else if (out_prec == PREC_BF16) {
      if (antialias) hipLaunchKernelGGL((crop_transform_kernel<true, __bf16>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((crop_transform_kernel<false, __bf16>), grid, dim3(256), 0, s, a);
    } else {
      if (antialias) hipLaunchKernelGGL((crop_transform_kernel<true, _Float16>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((crop_transform_kernel<false, _Float16>), grid, dim3(256), 0, s, a);
    }
    const int rc = check_launch("crop_transform");
    if (rc) return rc;
  }
  return EFFOCR_OK;
}

}  // namespace effocr
