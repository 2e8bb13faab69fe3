// Box stage of the ONNX driver's run_effocr (infer_effocr_onnx_multi.py:252-256,275-288,313-320; en/jp_preprocess :70-73,133-135) on the
// device, TWO launches for all lines of a call (rounds 3-5 ran ~25 ATen launches — sort, gathers, rounds, scatters — with host-side gaps
// between them: ~0.5 ms of an 18 ms call, profiles/r05_c5_call_kernels.txt):
//   boxes_sort_kernel     one workgroup per line: the localizer's NMS rows [max_det, 6] = (x0, y0, x1, y1, conf, label) of which the first
//                         counts[l] are valid; characters = label 0; STABLE sort of the characters along the reading axis (Python's
//                         sorted(bboxes_char, key=lambda x: x[axis]) — equal keys keep NMS order), everything else behind them in row order;
//                         -> sorted [max_det, 4] boxes, n_chars[l]
//   boxes_compact_kernel  crop slices of every character, compact over the lines in line order: torch.round(bbox) (half to even), scaled
//                         by size / 640 in float64 and rounded again (Python's round(): half to even; the literal 640 of :315-318), resolved
//                         like the numpy slice im[y0:y1, x0:x1] (negative bounds count from the end, clipped to [0, size]) and widened to
//                         the full line height (width when vertical) -> int32 [total, 5] = (x0, y0, x1, y1, line), total
// A bitonic network over (key, row) pairs in LDS: the row index as the second sort key makes it stable; NaN keys rank last (torch.sort).
#include "../../include/effocr_hip.h"
#include "common.hpp"
#include "kernels.hpp"

#include <math.h>

namespace effocr {
namespace {

constexpr int BOX_MAX_DET = 4096;

__device__ __forceinline__ bool pair_less(float ka, int ia, float kb, int ib) {
  const bool na = ka != ka, nb = kb != kb;
  if (na || nb) return na == nb ? ia < ib : nb;           // NaN after every number; two NaNs by row
  return ka < kb || (ka == kb && ia < ib);
}

__global__ __launch_bounds__(256) void boxes_sort_kernel(const float* __restrict__ rows, const int* __restrict__ counts, int max_det, int P, int axis,
                                                         float* __restrict__ sorted, int* __restrict__ n_chars) {
  extern __shared__ char box_smem[];
  float* key = reinterpret_cast<float*>(box_smem);
  int* idx = reinterpret_cast<int*>(box_smem) + P;
  __shared__ int nch;
  const int l = blockIdx.x, tid = threadIdx.x;
  const float* r = rows + (size_t)l * max_det * 6;
  int cnt = counts[l];
  cnt = cnt < 0 ? 0 : (cnt > max_det ? max_det : cnt);
  if (tid == 0) nch = 0;
  __syncthreads();
  int mine = 0;
  for (int j = tid; j < P; j += 256) {
    float k = INFINITY;
    if (j < cnt && r[j * 6 + 5] == 0.0f) { k = r[j * 6 + axis]; ++mine; }
    key[j] = k; idx[j] = j;                                // (rows past max_det: padding of the network, +inf, behind everything by index)
  }
  if (mine) atomicAdd(&nch, mine);
  __syncthreads();
  for (int k2 = 2; k2 <= P; k2 <<= 1) {
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      for (int t = tid; t < P; t += 256) {
        const int p = t ^ j2;
        if (p > t) {
          const float ka = key[t], kb = key[p];
          const int ia = idx[t], ib = idx[p];
          const bool up = (t & k2) == 0;
          const bool swap = up ? pair_less(kb, ib, ka, ia) : pair_less(ka, ia, kb, ib);
          if (swap) { key[t] = kb; key[p] = ka; idx[t] = ib; idx[p] = ia; }
        }
      }
      __syncthreads();
    }
  }
  float* o = sorted + (size_t)l * max_det * 4;
  for (int j = tid; j < max_det; j += 256) {
    const int s = idx[j];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[j * 4 + e] = r[s * 6 + e];
  }
  if (tid == 0) n_chars[l] = nch;
}

__device__ __forceinline__ int resolve_slice(long long v, int size) {
  if (v < 0) v += size;
  return (int)(v < 0 ? 0 : (v > size ? size : v));
}

__global__ __launch_bounds__(256) void boxes_compact_kernel(const float* __restrict__ sorted, const int* __restrict__ n_chars, int L, int max_det, int H, int W,
                                                            int vertical, int* __restrict__ boxes5, int* __restrict__ total) {
  const int l = blockIdx.x, tid = threadIdx.x;
  int off = 0, all = 0;
  for (int i = 0; i < L; ++i) {                           // (L is tens of lines: every workgroup adds up its own offset)
    const int n = n_chars[i];
    if (i < l) off += n;
    all += n;
  }
  if (l == 0 && tid == 0) *total = all;
  const int n = n_chars[l];
  const float* b = sorted + (size_t)l * max_det * 4;
  for (int j = tid; j < n; j += 256) {
    const double size = vertical ? (double)H : (double)W;
    const double lo_r = (double)rintf(b[j * 4 + (vertical ? 1 : 0)]), hi_r = (double)rintf(b[j * 4 + (vertical ? 3 : 2)]);   // torch.round on fp32, then .double()
    const long long lo = (long long)rint(lo_r * size / 640.0), hi = (long long)rint(hi_r * size / 640.0);
    const int a0 = resolve_slice(lo, vertical ? H : W), a1 = resolve_slice(hi, vertical ? H : W);
    int* o = boxes5 + (size_t)(off + j) * 5;
    if (vertical) { o[0] = 0; o[1] = a0; o[2] = W; o[3] = a1; }
    else { o[0] = a0; o[1] = 0; o[2] = a1; o[3] = H; }
    o[4] = l;
  }
}

}  // namespace
}  // namespace effocr

using namespace effocr;

extern "C" {

int effocr_parse_char_boxes(const float* rows_dev, const int* counts_dev, int lines, int max_det, int height, int width, int axis, int vertical,
                            float* sorted_dev, int* n_chars_dev, int* boxes5_dev, int* total_dev, void* stream) {
  if (lines < 0 || max_det <= 0 || max_det > BOX_MAX_DET) return fail(EFFOCR_EUNSUPPORTED, "parse_char_boxes: max_det must be in 1..4096");
  if (height <= 0 || width <= 0 || (axis != 0 && axis != 1)) return fail(EFFOCR_EINVAL, "parse_char_boxes: bad geometry / axis");
  if (!total_dev) return fail(EFFOCR_EINVAL, "parse_char_boxes: NULL output pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (lines == 0) return hipMemsetAsync(total_dev, 0, sizeof(int), s) == hipSuccess ? EFFOCR_OK : fail(EFFOCR_EHIP, "parse_char_boxes: memset failed");
  if (!rows_dev || !counts_dev || !sorted_dev || !n_chars_dev || !boxes5_dev) return fail(EFFOCR_EINVAL, "parse_char_boxes: NULL device pointer");
  int P = 64;
  while (P < max_det) P <<= 1;
  hipLaunchKernelGGL(boxes_sort_kernel, dim3((unsigned)lines), dim3(256), (size_t)P * 8, s, rows_dev, counts_dev, max_det, P, axis, sorted_dev, n_chars_dev);
  int rc = check_launch("boxes_sort");
  if (rc) return rc;
  hipLaunchKernelGGL(boxes_compact_kernel, dim3((unsigned)lines), dim3(256), 0, s, sorted_dev, n_chars_dev, lines, max_det, height, width, vertical, boxes5_dev, total_dev);
  return check_launch("boxes_compact");
}

}  // extern "C"
