// Non-GEMM kernels of the ViT recognizer encoder (timm VisionTransformer, called through
// models/encoders.py:58-64 at infer_effocr.py:314): LayerNorm, patch im2col, CLS row set-up,
// multi-head self-attention, and the final LayerNorm + CLS pooling (+ the F.normalize of
// infer_effocr.py:316 fused in).
#include "common.hpp"
#include "kernels.hpp"
#include "ln.hpp"
#include <math.h>


#ifndef ATT_NT
#define ATT_NT 0                                          // attention (token-panel path / ViT-B): non-temporal 1 = qkv loads, 2 = output stores (both measured slower: 3.65 -> 3.9 ms)
#endif
#ifndef LN_NT
#define LN_NT 1                                           // blocked LayerNorm: non-temporal 1 = row loads, 2 = stores
#endif
namespace effocr {
namespace {

template <int G, int V, typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t rows,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        TO* __restrict__ out) {
  constexpr int D = 4 * G * V;
  constexpr int RPW = 64 / G;
  const int lane = threadIdx.x & 63, sub = lane % G;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / G;
  const int64_t rc = row < rows ? row : rows - 1;
  f32x4 y[V];
  ln_row<G, V>(x + rc * D, sub, gamma, beta, eps, y);
  if (row < rows) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      TO* p = out + row * D + (sub + G * i) * 4;
      if constexpr (sizeof(TO) == 4) *reinterpret_cast<f32x4*>(p) = y[i];
      else *reinterpret_cast<u32x2*>(p) = pack4<TO>(y[i][0], y[i][1], y[i][2], y[i][3]);
    }
  }
}

// LayerNorm over the fragment-blocked layout: x fp32 cells [rows/32][D/4][32 rows][16 B] -> out 16-bit cells
// [rows/32][D/8][32 rows][16 B].  One workgroup per 32-row block; wave w owns a quarter of the columns, lane =
// (row tl = lane & 31, parity part = lane >> 5) holds the chunks w*D/16 + part + 2i in registers (every load is
// a contiguous 512-byte cell per half-wave).  Statistics: lane sums -> xor-32 exchange -> 4-wave exchange in
// LDS; exact two-pass variance from the registers (same arithmetic order class as ln_apply).  A lane with
// part = 0 / 1 holds the even / odd fp32 chunks = the low / high 8 bytes of one 16-byte output chunk.
// REDUCE (the split tail panels of the fused MLP, mlp_kernel.hpp): the row is first ASSEMBLED from the parts' fp32 partial outputs
// (fixed order) + bias2 (+ the old row), written back as the new residual row, then normalised — one launch instead of the
// reduction kernel + this one (each launch of a small-batch forward costs ~7 us + its queue gap); the arithmetic of both is unchanged.
struct LnReduce { const float* partial; const float* b2; float* xw; int nparts, tail_rb, add_x; };
// NP (REDUCE): parts per row as a compile-time constant (2 / 4: what the fused MLP's launcher cuts; 0 = rd.nparts at run time) — with it
// every partial-row load of a lane (12 chunks x NP parts at D = 384) is in flight at once; the run-time loop waited for each part in
// turn (20 us per launch on 160 workgroups at 1024 crops, 16 us at 64 crops: round 5).  Same fixed summation order.
template <int D, typename TO, bool REDUCE = false, int NP = 0>
__global__ __launch_bounds__(256) void layernorm_blocked_kernel(const float* __restrict__ x, int64_t rows,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps,
                                                                TO* __restrict__ out, LnReduce rd) {
  static_assert(sizeof(TO) == 2 && D % 32 == 0, "blocked LayerNorm writes 16-bit operands");
  constexpr int NQ = D / 32;                               // float4 chunks per lane
  __shared__ float red[2][4][32];
  const int lane = threadIdx.x & 63, tl = lane & 31, part = lane >> 5, wv = threadIdx.x >> 6;
  const int64_t rb = blockIdx.x;
  const char* xb = reinterpret_cast<const char*>(x) + rb * (D / 4) * 512 + tl * 16;
  const int c0 = wv * (D / 16) + part;
  f32x4 v[NQ];
  // (REDUCE — the launch between two kernels of a small call: the norm's weight / bias requested with the first loads instead of behind the
  // two statistics barriers, one exposed memory round trip less)
  f32x4 gmv[REDUCE ? NQ : 1], btv[REDUCE ? NQ : 1];
  if constexpr (REDUCE) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      gmv[i] = *reinterpret_cast<const f32x4*>(gamma + (c0 + 2 * i) * 4);
      btv[i] = *reinterpret_cast<const f32x4*>(beta + (c0 + 2 * i) * 4);
    }
  }
  if constexpr (REDUCE) {
    const int64_t per_rb = (int64_t)(D / 4) * 32;
    const bool live = rb * 32 + tl < rows;
    f32x4 pvs[NP > 0 ? NQ : 1][NP > 0 ? NP : 1];
    if constexpr (NP > 0) {
#pragma unroll
      for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p)
          pvs[i][p] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rd.partial) + ((int64_t)p * rd.tail_rb + rb) * per_rb + (int64_t)(c0 + 2 * i) * 32 + tl);
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int c = c0 + 2 * i;
      const int64_t rem = (int64_t)c * 32 + tl;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if constexpr (NP > 0) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += pvs[i][p][e];
      } else
      for (int p = 0; p < rd.nparts; ++p) {
        const f32x4 pv = reinterpret_cast<const f32x4*>(rd.partial)[((int64_t)p * rd.tail_rb + rb) * per_rb + rem];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += pv[e];
      }
      const f32x4 bv = *reinterpret_cast<const f32x4*>(rd.b2 + c * 4);
      f32x4 xo = {0.f, 0.f, 0.f, 0.f};
      if (rd.add_x) xo = *reinterpret_cast<const f32x4*>(xb + (size_t)c * 512);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = acc[e] + bv[e] + (rd.add_x ? xo[e] : 0.f);
      if (live) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(rd.xw) + rb * (D / 4) * 512 + tl * 16 + (size_t)c * 512) = v[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < NQ; ++i) v[i] = (LN_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xb + (size_t)(c0 + 2 * i) * 512)) : *reinterpret_cast<const f32x4*>(xb + (size_t)(c0 + 2 * i) * 512);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NQ; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  s += __shfl_xor(s, 32, 64);
  if (part == 0) red[0][wv][tl] = s;
  __syncthreads();
  const float mean = ((red[0][0][tl] + red[0][1][tl]) + (red[0][2][tl] + red[0][3][tl])) * (1.0f / D);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NQ; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; ss += d * d; }
  ss += __shfl_xor(ss, 32, 64);
  if (part == 0) red[1][wv][tl] = ss;
  __syncthreads();
  const float var = ((red[1][0][tl] + red[1][1][tl]) + (red[1][2][tl] + red[1][3][tl])) * (1.0f / D);
  const float rstd = 1.0f / sqrtf(var + eps);
  if (rb * 32 + tl >= rows) return;                        // padding rows of the last block: nothing to write
  char* ob = reinterpret_cast<char*>(out) + rb * (D / 8) * 512 + tl * 16 + part * 8;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int c = c0 + 2 * i;
    const f32x4 gm = REDUCE ? gmv[REDUCE ? i : 0] : *reinterpret_cast<const f32x4*>(gamma + c * 4);
    const f32x4 bt = REDUCE ? btv[REDUCE ? i : 0] : *reinterpret_cast<const f32x4*>(beta + c * 4);
    const u32x2 o = pack4<TO>((v[i][0] - mean) * rstd * gm[0] + bt[0], (v[i][1] - mean) * rstd * gm[1] + bt[1],
                              (v[i][2] - mean) * rstd * gm[2] + bt[2], (v[i][3] - mean) * rstd * gm[3] + bt[3]);
    if (LN_NT & 2) __builtin_nontemporal_store(o, reinterpret_cast<u32x2*>(ob + (size_t)(c >> 1) * 512));
    else *reinterpret_cast<u32x2*>(ob + (size_t)(c >> 1) * 512) = o;
  }
}

// final LayerNorm on the CLS row of every image (+ optional L2 normalisation) -> emb [B,D] fp32
template <int G, int V>
__global__ __launch_bounds__(256) void cls_norm_kernel(const float* __restrict__ x, int B, int T,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int l2norm,
                                                       int blocked, float* __restrict__ emb, int* __restrict__ status) {
  constexpr int D = 4 * G * V;
  constexpr int RPW = 64 / G;
  const int lane = threadIdx.x & 63, sub = lane % G;
  const int img = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / G;
  const int ic = img < B ? img : B - 1;
  f32x4 y[V], v[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int chunk = sub + G * i;
    const char* p = blocked ? reinterpret_cast<const char*>(x) + blk_off((int64_t)ic * T, chunk, D / 4)
                            : reinterpret_cast<const char*>(x + (int64_t)ic * T * D + chunk * 4);
    v[i] = *reinterpret_cast<const f32x4*>(p);
  }
  ln_apply<G, V>(v, sub, gamma, beta, eps, y);
  if (l2norm) {
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) ss += y[i][e] * y[i][e];
    const float nrm = fmaxf(sqrtf(group_sum<G>(ss)), 1e-12f);    // F.normalize: x / max(||x||, eps)
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) y[i][e] = y[i][e] / nrm;
  }
  if (img < B) {
    // status word (effocr_encoder_check_status): a non-finite embedding = an inf / nan reached the class-token row — a 16-bit operand
    // overflowed on the way (f16: |q|, |k|, |v|, |fc1 pre-activation| > 65504 rounds to inf, and an inf anywhere in a block turns the
    // next LayerNorm / softmax of every row it feeds into nan) or the input itself was not finite.
    bool bad = false;
#pragma unroll
    for (int i = 0; i < V; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bad |= !(fabsf(y[i][e]) <= 3.0e38f);
      *reinterpret_cast<f32x4*>(emb + (int64_t)img * D + (sub + G * i) * 4) = y[i];
    }
    if (status && bad) atomicOr(status, 1);
  }
}

// ------------------------------------------------------------------------------------------
// im2col for the 16x16/stride-16 patch-embedding conv: x [B,3,H,W] fp32 NCHW ->
// rows [(img,py,px)][k = c*256 + ky*16 + kx] in the GEMM operand type (k order = the conv weight's
// own [D,3,16,16] flattening, so the weight needs no permutation).
// ------------------------------------------------------------------------------------------
template <typename TO, typename TI = float>               // TI = TO (16-bit): the crops arrive in the operand type already — a plain gather
__global__ __launch_bounds__(256) void im2col16_kernel(const TI* __restrict__ x, int B, int H, int W,
                                                       TO* __restrict__ out) {
  const int PH = H / 16, PW = W / 16;
  const int64_t total = (int64_t)B * PH * PW * 96;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= total) return;
  const int k8 = (int)(id % 96);
  const int64_t m = id / 96;
  const int px = (int)(m % PW), py = (int)((m / PW) % PH);
  const int64_t img = m / ((int64_t)PW * PH);
  const int c = k8 >> 5, ky = (k8 & 31) >> 1, kx0 = (k8 & 1) * 8;
  const TI* src = x + ((img * 3 + c) * H + (py * 16 + ky)) * (int64_t)W + px * 16 + kx0;
  TO* dst = out + m * 768 + k8 * 8;
  if constexpr (sizeof(TI) == 2) {
    static_assert(sizeof(TO) == 2, "im2col16: 16-bit crops feed 16-bit patch rows");
    *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
    return;
  }
  const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src));
  const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + 4);
  if constexpr (sizeof(TO) == 4) {
    *reinterpret_cast<f32x4*>(dst) = a;
    *reinterpret_cast<f32x4*>(dst + 4) = b;
  } else {
    const u32x2 lo = pack4<TO>(a[0], a[1], a[2], a[3]);
    const u32x2 hi = pack4<TO>(b[0], b[1], b[2], b[3]);
    u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    *reinterpret_cast<u32x4*>(dst) = v;
  }
}

// Last transformer block: only the class-token row of every image reaches the output (global_pool = 'token'), and
// attn.proj / the MLP / both LayerNorms act on each row independently — so the block's second half runs on B gathered rows.
// Gathers row img*T of x (fp32 blocked) and of the attention output (16-bit blocked) into compact blocked buffers [B rows].
__global__ __launch_bounds__(256) void gather_cls_kernel(const float* __restrict__ x, const char* __restrict__ att, int B, int T, int D,
                                                         float* __restrict__ xc, char* __restrict__ ac) {
  const int nx = D / 4, na = D / 8;                       // 16-byte chunks per row
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (int64_t)B * (nx + na)) return;
  const int img = (int)(id / (nx + na)), c = (int)(id % (nx + na));
  if (c < nx) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(xc) + blk_off(img, c, nx)) =
                  *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(x) + blk_off((int64_t)img * T, c, nx));
  else *reinterpret_cast<u32x4*>(ac + blk_off(img, c - nx, na)) = *reinterpret_cast<const u32x4*>(att + blk_off((int64_t)img * T, c - nx, na));
}

// token 0 of every image = cls_token + pos_embed[0] (pre-added on the host at weight upload)
__global__ void set_cls_kernel(const float* __restrict__ cls_pos0, float* __restrict__ x, int B, int T, int D, int blocked, int* __restrict__ status_zero) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id == 0 && status_zero) *status_zero = 0;          // first kernel of a forward: the status word starts clean
  if (id >= (int64_t)B * D) return;
  const int d = (int)(id % D);
  const int64_t img = id / D;
  if (blocked) *reinterpret_cast<float*>(reinterpret_cast<char*>(x) + blk_off(img * T, d >> 2, D / 4) + (d & 3) * 4) = cls_pos0[d];
  else x[img * T * D + d] = cls_pos0[d];
}

// ------------------------------------------------------------------------------------------
// Multi-head self-attention, head_dim 64, whole sequence (T <= 32*NKT keys) in one workgroup.
//   qkv [B*T, 3*D] (q | k | v, feature = which*D + h*64 + d, timm's reshape(B,N,3,H,hd)),
//   out [B*T, D]   (feature = h*64 + d)  ==  (softmax(q k^T / 8) v).transpose(1,2).reshape(B,N,D)
// One workgroup (4 waves) per (image, head).  K (row-major, 144-B padded rows) and V^T
// (keys contiguous, packed key pairs) are staged once in LDS; each wave then owns 32-query
// blocks.  Scores are computed SWAPPED, S^T = K Q^T with v_mfma_f32_32x32x16, so that a lane
// holds one query column and all of its keys in registers: the softmax max / sum are lane-local
// plus ONE cross-half exchange, no online rescaling (the whole row is resident, <= 112
// accumulator registers).  The un-normalised P fragment that falls out of the S^T C-layout is
// already a valid B-operand for O^T = V^T P^T provided V^T is read with the matching key
// permutation (keys {0-3,8-11 | 4-7,12-15} + 16m per half-wave), so no permute instructions.
// ------------------------------------------------------------------------------------------
// BLK: qkv and out are fragment-blocked (see blk_off): row = global token index, 16-B chunk = 8 features.
template <typename E, int NKT, bool BLK>
__global__ __launch_bounds__(256, 2) void attn_mfma_kernel(const E* __restrict__ qkv, E* __restrict__ out,
                                                           int B, int T, int heads) {
  typedef typename Op16<E>::V8 V8;
  constexpr int TP = 32 * NKT;
  constexpr int KROW = 144;                 // bytes per K row: 64 elements + 16 B pad
  constexpr int VS = TP / 2 + 6;            // dwords per V^T row (even, VS/2 odd -> conflict-free b64 reads)
  __shared__ __attribute__((aligned(16))) char smem[TP * KROW + 64 * VS * 4];
  char* sK = smem;
  uint32_t* sV = reinterpret_cast<uint32_t*>(smem + TP * KROW);

  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int w = wave_id();
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int D = heads * 64;
  const int64_t ld = 3 * (int64_t)D;
  const int64_t tok0 = (int64_t)b * T;                   // first token of this image
  const int nch = 3 * D / 8;                             // chunks per qkv row
  // 16-B chunk `c8` (0..7) of section `sec` (0 q, 1 k, 2 v) of token t
  auto qkv_ptr = [&](int t, int sec, int c8) -> const u32x4* {
    if constexpr (BLK) return reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(qkv) + blk_off(tok0 + t, (sec * D + h * 64) / 8 + c8, nch));
    else return reinterpret_cast<const u32x4*>(qkv + (tok0 + t) * ld + sec * D + h * 64 + c8 * 8);
  };

  auto ld_qkv = [&](const u32x4* p) -> u32x4 { return (ATT_NT & 1) ? __builtin_nontemporal_load(p) : *p; };   // (q, k, v rows are read once)
  auto load_q = [&](V8 (&q)[4], int qb) {
    int tq = qb * 32 + r31;
    tq = tq < T ? tq : T - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) q[ks] = __builtin_bit_cast(V8, ld_qkv(qkv_ptr(tq, 0, 2 * ks + half)));
  };
  // K rows (zero rows beyond T) and V -> registers.  Every global load of the workgroup is issued before the
  // first LDS write: ONE exposed memory latency per workgroup instead of one per loop iteration (2.50 -> 2.01 ms;
  // prefetching the next head's K / V across a multi-head loop was tried and spills: the 224-key score row
  // already occupies 112 registers)
  constexpr int NKI = TP * 8 / 256;                         // K: 16-B chunks per thread
  constexpr int NVI = ((TP / 2) * 8 + 255) / 256;           // V: key-pair chunks per thread
  u32x4 kreg[NKI], v0reg[NVI], v1reg[NVI];
  auto load_kv = [&]() {
#pragma unroll
    for (int i = 0; i < NKI; ++i) {
      const int id = tid + 256 * i, t = id >> 3, c = id & 7;
      const u32x4 z = {0u, 0u, 0u, 0u};
      kreg[i] = z;
      if (t < T) kreg[i] = ld_qkv(qkv_ptr(t, 1, c));
    }
#pragma unroll
    for (int i = 0; i < NVI; ++i) {
      const int id = tid + 256 * i, tp = id >> 3, c = id & 7, t0 = 2 * tp;
      const u32x4 z = {0u, 0u, 0u, 0u};
      v0reg[i] = z; v1reg[i] = z;
      if (t0 < T) v0reg[i] = ld_qkv(qkv_ptr(t0, 2, c));
      if (t0 + 1 < T) v1reg[i] = ld_qkv(qkv_ptr(t0 + 1, 2, c));
    }
  };
  // registers -> LDS: K row-major, V transposed: dword (d, tp) = {V[2tp][d], V[2tp+1][d]}
  auto store_kv = [&]() {
#pragma unroll
    for (int i = 0; i < NKI; ++i) {
      const int id = tid + 256 * i, t = id >> 3, c = id & 7;
      *reinterpret_cast<u32x4*>(sK + t * KROW + c * 16) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < NVI; ++i) {
      const int id = tid + 256 * i, tp = id >> 3, c = id & 7;
      if (id < (TP / 2) * 8) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const uint32_t a = v0reg[i][jj], bq = v1reg[i][jj];
          sV[(c * 8 + 2 * jj) * VS + tp] = (a & 0xffffu) | (bq << 16);
          sV[(c * 8 + 2 * jj + 1) * VS + tp] = (a >> 16) | (bq & 0xffff0000u);
        }
      }
    }
  };

  V8 qf[4], qn[4];
  if (w * 32 < T) load_q(qf, w);                            // oldest in the VM queue: ready when the staging is
  load_kv();
  store_kv();
  __syncthreads();

  const float cexp = 0.125f * 1.44269504088896340736f;     // head_dim^-0.5 * log2(e)
  for (int qb = w; qb * 32 < T; qb += 4) {
    int tq = qb * 32 + r31;
    const bool qvalid = tq < T;
    tq = qvalid ? tq : T - 1;
    const bool more = (qb + 4) * 32 < T;
    if (more) load_q(qn, qb + 4);

    // S^T tiles: rows = keys, cols = queries
    f32x16 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const V8 kf = *reinterpret_cast<const V8*>(sK + (kt * 32 + r31) * KROW + (2 * ks + half) * 16);
        s[kt] = Op16<E>::mfma(kf, qf[ks], s[kt]);
      }
    }
    // mask the padded keys of the last tile, row max
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (kt == NKT - 1) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= T) s[kt][r] = -INFINITY;
        }
        mx = fmaxf(mx, s[kt][r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // P = exp2((S - max) * c) tile by tile, fed straight into O^T = V^T P^T (rows = head dims,
    // cols = queries) so that each score tile's registers die as soon as it is consumed
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float l = 0.f;
    const float mxc = mx * cexp;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        V8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][8 * m + j], cexp, -mxc));
          l += p;
          pf[j] = (E)p;
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const uint32_t* vp = sV + (db * 32 + r31) * VS + (kt * 16 + 8 * m + 2 * half);
          const u32x2 lo = *reinterpret_cast<const u32x2*>(vp);
          const u32x2 hi = *reinterpret_cast<const u32x2*>(vp + 4);
          const u32x4 vv = {lo[0], lo[1], hi[0], hi[1]};
          o[db] = Op16<E>::mfma(__builtin_bit_cast(V8, vv), pf, o[db]);
        }
      }
      // keep the scheduler from hoisting every tile's exp / V^T reads to the top (424 live registers)
      __builtin_amdgcn_sched_barrier(0);
    }
    l += __shfl_xor(l, 32, 64);
    if (qvalid) {
      const float inv = 1.0f / l;
      E* orow = out + (tok0 + tq) * D + h * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int d = db * 32 + 8 * q4 + 4 * half;
          u32x2* dst = reinterpret_cast<u32x2*>(orow + d);
          if constexpr (BLK) dst = reinterpret_cast<u32x2*>(reinterpret_cast<char*>(out) + blk_off(tok0 + tq, (h * 64 + db * 32 + 8 * q4) / 8, D / 8) + half * 8);
          const u32x2 ov = pack4<E>(o[db][4 * q4] * inv, o[db][4 * q4 + 1] * inv, o[db][4 * q4 + 2] * inv, o[db][4 * q4 + 3] * inv);
          if (ATT_NT & 2) __builtin_nontemporal_store(ov, dst); else *dst = ov;
        }
    }
    if (more) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
    }
  }
}

// fp32 parity mode: straightforward one-thread-per-query attention with K, V in LDS (broadcast
// reads), online softmax in fp32.  Correctness reference path, not a throughput kernel.
constexpr int ATT32_TMAX = 224;
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                       int B, int T, int heads) {
  __shared__ __attribute__((aligned(16))) float sK[ATT32_TMAX * 64];
  __shared__ __attribute__((aligned(16))) float sV[ATT32_TMAX * 64];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int D = heads * 64;
  const int64_t ld = 3 * (int64_t)D;
  const float* base = qkv + (int64_t)b * T * ld + h * 64;
  for (int id = tid; id < T * 16; id += 256) {
    const int t = id >> 4, c = id & 15;
    *reinterpret_cast<f32x4*>(sK + t * 64 + c * 4) = *reinterpret_cast<const f32x4*>(base + (int64_t)t * ld + D + c * 4);
    *reinterpret_cast<f32x4*>(sV + t * 64 + c * 4) = *reinterpret_cast<const f32x4*>(base + (int64_t)t * ld + 2 * D + c * 4);
  }
  __syncthreads();
  for (int tq = tid; tq < T; tq += 256) {
    float q[64], o[64];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)tq * ld + c * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { q[c * 4 + e] = v[e] * 0.125f; o[c * 4 + e] = 0.f; }
    }
    float mx = -INFINITY, l = 0.f;
    for (int key = 0; key < T; ++key) {
      float sc = 0.f;
#pragma unroll
      for (int d = 0; d < 64; ++d) sc = fmaf(q[d], sK[key * 64 + d], sc);
      const float mn = fmaxf(mx, sc);
      const float alpha = expf(mx - mn);
      const float p = expf(sc - mn);
      l = l * alpha + p;
#pragma unroll
      for (int d = 0; d < 64; ++d) o[d] = fmaf(p, sV[key * 64 + d], o[d] * alpha);
      mx = mn;
    }
    const float inv = 1.0f / l;
    float* orow = out + ((int64_t)b * T + tq) * D + h * 64;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      f32x4 v = {o[c * 4] * inv, o[c * 4 + 1] * inv, o[c * 4 + 2] * inv, o[c * 4 + 3] * inv};
      *reinterpret_cast<f32x4*>(orow + c * 4) = v;
    }
  }
}

template <typename TO>
int launch_ln(const float* x, int64_t rows, int D, const float* gamma, const float* beta, float eps, TO* out,
              hipStream_t s) {
  if (rows <= 0) return EFFOCR_OK;
  if (D == 384) {
    const int64_t rpb = 4 * 2;
    hipLaunchKernelGGL((layernorm_kernel<32, 3, TO>), dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, s, x, rows, gamma, beta, eps, out);
  } else if (D == 768) {
    const int64_t rpb = 4;
    hipLaunchKernelGGL((layernorm_kernel<64, 3, TO>), dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, s, x, rows, gamma, beta, eps, out);
  } else if (D == 128) {
    const int64_t rpb = 4 * 2;
    hipLaunchKernelGGL((layernorm_kernel<32, 1, TO>), dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, s, x, rows, gamma, beta, eps, out);
  } else {
    return fail(EFFOCR_EUNSUPPORTED, "layernorm: embed dim must be 128, 384 or 768");
  }
  return check_launch("layernorm");
}

template <typename TO>
int launch_ln_blocked(const float* x, int64_t rows, int D, const float* gamma, const float* beta, float eps, TO* out,
                      hipStream_t s) {
  if (rows <= 0) return EFFOCR_OK;
  const dim3 grid((unsigned)((rows + 31) / 32));
  const LnReduce none{};
  if (D == 768) hipLaunchKernelGGL((layernorm_blocked_kernel<768, TO>), grid, dim3(256), 0, s, x, rows, gamma, beta, eps, out, none);
  else if (D == 384) hipLaunchKernelGGL((layernorm_blocked_kernel<384, TO>), grid, dim3(256), 0, s, x, rows, gamma, beta, eps, out, none);
  else if (D == 128) hipLaunchKernelGGL((layernorm_blocked_kernel<128, TO>), grid, dim3(256), 0, s, x, rows, gamma, beta, eps, out, none);
  else return fail(EFFOCR_EUNSUPPORTED, "layernorm(blocked): embed dim must be 128, 384 or 768");
  return check_launch("layernorm_blocked");
}

template <typename TO, typename TI = float>
int launch_im2col(const void* x, int B, int H, int W, TO* out, hipStream_t s) {
  const int64_t total = (int64_t)B * (H / 16) * (W / 16) * 96;
  if (total <= 0) return EFFOCR_OK;
  hipLaunchKernelGGL((im2col16_kernel<TO, TI>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, static_cast<const TI*>(x), B, H, W, out);
  return check_launch("im2col16");
}

template <typename E, bool BLK>
int launch_attn_mfma(const E* qkv, E* out, int B, int T, int heads, hipStream_t s) {
  const int nkt = (T + 31) / 32;
  const dim3 grid((unsigned)(B * heads)), blk(256);
  switch (nkt) {
    case 1: hipLaunchKernelGGL((attn_mfma_kernel<E, 1, BLK>), grid, blk, 0, s, qkv, out, B, T, heads); break;
    case 2: hipLaunchKernelGGL((attn_mfma_kernel<E, 2, BLK>), grid, blk, 0, s, qkv, out, B, T, heads); break;
    case 7: hipLaunchKernelGGL((attn_mfma_kernel<E, 7, BLK>), grid, blk, 0, s, qkv, out, B, T, heads); break;
    default: return fail(EFFOCR_EUNSUPPORTED, "attention: token count must be <=64 or in (192,224]");
  }
  return check_launch("attention");
}

}  // namespace

int gather_cls_rows_blocked(const float* x, const void* att, int B, int T, int D, float* xc, void* ac, hipStream_t s) {
  if (B <= 0) return EFFOCR_OK;
  const int64_t total = (int64_t)B * (D / 4 + D / 8);
  hipLaunchKernelGGL(gather_cls_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, static_cast<const char*>(att), B, T, D, xc,
                     static_cast<char*>(ac));
  return check_launch("gather_cls");
}

int layernorm_rows(int prec_out, const float* x, int64_t rows, int D, const float* gamma, const float* beta,
                   float eps, void* out, hipStream_t s) {
  switch (prec_out) {
    case PREC_BF16: return launch_ln<__bf16>(x, rows, D, gamma, beta, eps, static_cast<__bf16*>(out), s);
    case PREC_FP16: return launch_ln<_Float16>(x, rows, D, gamma, beta, eps, static_cast<_Float16*>(out), s);
    case PREC_FP32: return launch_ln<float>(x, rows, D, gamma, beta, eps, static_cast<float*>(out), s);
  }
  return fail(EFFOCR_EINVAL, "layernorm: unknown precision");
}

// x and out fragment-blocked (common.hpp blk_off); the x buffer must be addressable up to the next multiple of 32 rows
template <typename TO>
int launch_reduce_ln_blocked(float* x, int64_t rows, int D, const float* partial, const float* b2, int nparts, int tail_rb, int add_x,
                             const float* gamma, const float* beta, float eps, TO* out, hipStream_t s) {
  if (rows <= 0) return EFFOCR_OK;
  const dim3 grid((unsigned)((rows + 31) / 32));
  const LnReduce rd{partial, b2, x, nparts, tail_rb, add_x};
  if (D == 384 && nparts == 6) hipLaunchKernelGGL((layernorm_blocked_kernel<384, TO, true, 6>), grid, dim3(256), 0, s, x, rows, gamma, beta, eps, out, rd);
  else if (D == 384 && nparts == 3) hipLaunchKernelGGL((layernorm_blocked_kernel<384, TO, true, 3>), grid, dim3(256), 0, s, x, rows, gamma, beta, eps, out, rd);
  else if (D == 384 && nparts == 4) hipLaunchKernelGGL((layernorm_blocked_kernel<384, TO, true, 4>), grid, dim3(256), 0, s, x, rows, gamma, beta, eps, out, rd);
  else if (D == 384 && nparts == 2) hipLaunchKernelGGL((layernorm_blocked_kernel<384, TO, true, 2>), grid, dim3(256), 0, s, x, rows, gamma, beta, eps, out, rd);
  else if (D == 384) hipLaunchKernelGGL((layernorm_blocked_kernel<384, TO, true>), grid, dim3(256), 0, s, x, rows, gamma, beta, eps, out, rd);
  else if (D == 128) hipLaunchKernelGGL((layernorm_blocked_kernel<128, TO, true>), grid, dim3(256), 0, s, x, rows, gamma, beta, eps, out, rd);
  else return fail(EFFOCR_EUNSUPPORTED, "reduce + layernorm(blocked): embed dim must be 128 or 384");
  return check_launch("reduce_layernorm_blocked");
}

// x (row block 0 = the first split panel) <- sum of the parts' partial outputs + bias2 (+ x), then its LayerNorm -> out, one launch
int reduce_layernorm_rows_blocked(int prec_out, float* x, int64_t rows, int D, const float* partial, const float* b2, int nparts, int tail_rb,
                                  int add_x, const float* gamma, const float* beta, float eps, void* out, hipStream_t s) {
  switch (prec_out) {
    case PREC_BF16: return launch_reduce_ln_blocked<__bf16>(x, rows, D, partial, b2, nparts, tail_rb, add_x, gamma, beta, eps, static_cast<__bf16*>(out), s);
    case PREC_FP16: return launch_reduce_ln_blocked<_Float16>(x, rows, D, partial, b2, nparts, tail_rb, add_x, gamma, beta, eps, static_cast<_Float16*>(out), s);
  }
  return fail(EFFOCR_EUNSUPPORTED, "reduce + layernorm(blocked): 16-bit output only");
}

int layernorm_rows_blocked(int prec_out, const float* x, int64_t rows, int D, const float* gamma, const float* beta,
                           float eps, void* out, hipStream_t s) {
  switch (prec_out) {
    case PREC_BF16: return launch_ln_blocked<__bf16>(x, rows, D, gamma, beta, eps, static_cast<__bf16*>(out), s);
    case PREC_FP16: return launch_ln_blocked<_Float16>(x, rows, D, gamma, beta, eps, static_cast<_Float16*>(out), s);
  }
  return fail(EFFOCR_EUNSUPPORTED, "layernorm(blocked): 16-bit output only");
}

int im2col_patch16(int prec_out, const void* x, int x16, int B, int H, int W, void* out, hipStream_t s) {
  if (H % 16 || W % 16) return fail(EFFOCR_EINVAL, "im2col: image size must be a multiple of 16");
  if (x16) {
    if (prec_out == PREC_BF16) return launch_im2col<__bf16, __bf16>(x, B, H, W, static_cast<__bf16*>(out), s);
    if (prec_out == PREC_FP16) return launch_im2col<_Float16, _Float16>(x, B, H, W, static_cast<_Float16*>(out), s);
    return fail(EFFOCR_EUNSUPPORTED, "im2col: 16-bit crops need a 16-bit precision mode");
  }
  switch (prec_out) {
    case PREC_BF16: return launch_im2col<__bf16>(x, B, H, W, static_cast<__bf16*>(out), s);
    case PREC_FP16: return launch_im2col<_Float16>(x, B, H, W, static_cast<_Float16*>(out), s);
    case PREC_FP32: return launch_im2col<float>(x, B, H, W, static_cast<float*>(out), s);
  }
  return fail(EFFOCR_EINVAL, "im2col: unknown precision");
}

int set_cls_rows(const float* cls_pos0, float* x, int B, int T, int D, int blocked, int* status_zero, hipStream_t s) {
  const int64_t total = (int64_t)B * D;
  if (total <= 0) return EFFOCR_OK;
  hipLaunchKernelGGL(set_cls_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cls_pos0, x, B, T, D, blocked, status_zero);
  return check_launch("set_cls_rows");
}

int attention(int prec, const void* qkv, void* out, int B, int T, int heads, int blocked, hipStream_t s) {
  if (B <= 0) return EFFOCR_OK;
  if (blocked && prec == PREC_FP32) return fail(EFFOCR_EUNSUPPORTED, "attention(fp32): blocked layout not supported");
  switch (prec) {
    case PREC_BF16:
      return blocked ? launch_attn_mfma<__bf16, true>(static_cast<const __bf16*>(qkv), static_cast<__bf16*>(out), B, T, heads, s)
                     : launch_attn_mfma<__bf16, false>(static_cast<const __bf16*>(qkv), static_cast<__bf16*>(out), B, T, heads, s);
    case PREC_FP16:
      return blocked ? launch_attn_mfma<_Float16, true>(static_cast<const _Float16*>(qkv), static_cast<_Float16*>(out), B, T, heads, s)
                     : launch_attn_mfma<_Float16, false>(static_cast<const _Float16*>(qkv), static_cast<_Float16*>(out), B, T, heads, s);
    case PREC_FP32:
      if (T > ATT32_TMAX) return fail(EFFOCR_EUNSUPPORTED, "attention(fp32): more than 224 tokens");
      hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)(B * heads)), dim3(256), 0, s,
                         static_cast<const float*>(qkv), static_cast<float*>(out), B, T, heads);
      return check_launch("attention_f32");
  }
  return fail(EFFOCR_EINVAL, "attention: unknown precision");
}

// (s_memtime ticks at the shader clock on gfx950, s_memrealtime at the constant 100 MHz reference: two samples give the average
// shader clock over the interval between them — bench.py brackets its timed region with it)
// s_memtime counts shader clocks PER CU (256 counters with unrelated offsets, tools/ubench/memtime_domain.hip): 1024 one-wave workgroups
// cover the CUs, each stores { s_memtime, s_memrealtime (100 MHz, chip-wide) } into the pair of ITS CU — key = XCC_ID * 256 + (SE, SH, CU)
// of HW_ID, 2048 pairs — so that two samples compare like with like.
__global__ void clock_sample_kernel(unsigned long long* out) {
  const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));          // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u;    // HW_REG_XCC_ID[3:0]
  const unsigned key = xcc * 256u + ((hw >> 13) & 7u) * 32u + ((hw >> 12) & 1u) * 16u + ((hw >> 8) & 15u);
  if (threadIdx.x == 0) { out[2 * key] = __builtin_amdgcn_s_memtime(); out[2 * key + 1] = __builtin_amdgcn_s_memrealtime(); }
}
int clock_sample(unsigned long long* out, hipStream_t s) {
  hipLaunchKernelGGL(clock_sample_kernel, dim3(1024), dim3(64), 0, s, out);
  return check_launch("clock_sample");
}

int final_cls_norm(const float* x, int B, int T, int D, const float* gamma, const float* beta, float eps,
                   int l2norm, int blocked, float* emb, int* status, hipStream_t s) {
  if (B <= 0) return EFFOCR_OK;
  if (D == 384) {
    hipLaunchKernelGGL((cls_norm_kernel<32, 3>), dim3((unsigned)((B + 7) / 8)), dim3(256), 0, s, x, B, T, gamma, beta, eps, l2norm, blocked, emb, status);
  } else if (D == 768) {
    hipLaunchKernelGGL((cls_norm_kernel<64, 3>), dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, x, B, T, gamma, beta, eps, l2norm, blocked, emb, status);
  } else if (D == 128) {
    hipLaunchKernelGGL((cls_norm_kernel<32, 1>), dim3((unsigned)((B + 7) / 8)), dim3(256), 0, s, x, B, T, gamma, beta, eps, l2norm, blocked, emb, status);
  } else {
    return fail(EFFOCR_EUNSUPPORTED, "final norm: embed dim must be 128, 384 or 768");
  }
  return check_launch("final_cls_norm");
}

}  // namespace effocr
