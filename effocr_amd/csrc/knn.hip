// Exact inner-product k-NN against the reference-glyph index: the faiss.IndexFlatIP.search role.
//
// Reference call sites: infer_effocr.py:184-187,317 (FaissKNN(index_init_fn=faiss.IndexFlatIP),
// knn_func(emb, k=10)); infer_effocr_onnx_multi.py:372 (k=1); train_effocr_recognizer.py:47-52
// (index rows = L2-normalised embeddings).  Semantics: S = Q.X^T in fp32, per query the k largest
// scores in descending order with their row ids; k > ntotal pads (score -FLT_MAX, id -1).
//
// Design for gfx950:
//   * S is never materialised.  A workgroup owns 128 queries x a contiguous chunk of index rows and
//     walks the chunk in 128-row tiles with the shared fp32 MFMA tile pipeline (tile128.hpp):
//     v_mfma_f32_32x32x2_f32, exact fp32, k ascending — so every score is bit-for-bit the
//     ascending-k fmaf chain the C oracle (oracle/flat_ip.c) computes, and ids match exactly even
//     at near-ties.  No split-K (it would break the chain).
//   * the MFMA is issued swapped (rows = index entries, cols = queries): a lane's 16 accumulators
//     of a 32x32 tile all belong to ONE query, index id ascending with the register number, so the
//     running top-k is a lane-local sorted list in registers (one threshold compare per score in
//     the common case; ties keep the earlier = lower id).
//   * per workgroup the 4 partial lists of a query (2 index-half waves x 2 half-waves) are merged
//     through LDS; per-chunk partial results go to the workspace and a second tiny kernel merges
//     the chunks (skipped when there is a single chunk).
// Tie rule (defined by this implementation, faiss leaves it unspecified): equal scores rank by
// ascending row id.
#include "common.hpp"
#include "kernels.hpp"
#include "tile128.hpp"
#include <float.h>
#include <limits.h>
#include <type_traits>

namespace effocr {
namespace {

using namespace tile128;

int g_knn_wg_target = 1024;          // workgroups a launch aims for (2 resident per CU); knn_set_option("wg_target")
bool g_knn_force_tile = false;        // A/B switch (tests): 1 = always the 128-query tile kernel
bool g_knn_q16 = true;                // A/B switch (tests): 0 = calls of <= 16 queries on the 32-wide tile as well
int g_knn_stream_min_rows = 65536;    // 33..128 queries: index rows from which the streaming kernel (not the tile kernel) runs; knn_set_option("stream_min_rows")
bool g_knn_two_pass = false;          // A/B switch (tests): 1 = screened search collects its candidates with a second scan of the index
constexpr int ID_NONE = INT_MAX;          // internal sentinel id (ranks after every real id)
constexpr int MAX_CHUNKS = 256;

__device__ __forceinline__ bool before(float s1, int i1, float s2, int i2) {
  return (s1 > s2) || (s1 == s2 && i1 < i2);
}

// ---- wave-wide "best of 64" in the compound order (score desc, id asc), on the DPP cross-lane paths --------------------------------
// (the ds_bpermute form — two LDS-crossbar round trips per step, six steps, once per output — was most of the merge kernels' time).
// Scores map to order-preserving unsigned keys (-0 folded onto +0 so that key equality is float equality); the winner is the lane with
// the largest key and, among equal keys, the smallest id: one max-reduction and one min-reduction, 6 DPP steps each, result in lane 63.
// A NaN score (a non-finite query: e.g. an f16 operand overflow in the encoder) maps to key 0, below every real score AND below the
// (-FLT_MAX, ID_NONE) sentinel: it can never win a merge, so such a query comes back as (-FLT_MAX, -1) padding, never as a plausible id.
__device__ __forceinline__ unsigned f32_key(float f) {
  unsigned u = __float_as_uint(f);
  u = (u == 0x80000000u) ? 0u : u;
  const unsigned key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return (f != f) ? 0u : key;
}
__device__ __forceinline__ float key_f32(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
template <bool MAX> __device__ __forceinline__ unsigned wave_reduce_u32(unsigned v) {
  constexpr int ident = MAX ? 0 : -1;                       // lanes without a source (row starts / unselected rows) contribute the identity
#define KNN_DPP_STEP(ctrl, rows)                                                                            \
  {                                                                                                         \
    const unsigned t_ = (unsigned)__builtin_amdgcn_update_dpp(ident, (int)v, ctrl, rows, 0xf, false);       \
    v = MAX ? (v > t_ ? v : t_) : (v < t_ ? v : t_);                                                        \
  }
  KNN_DPP_STEP(0x111, 0xf)                                   // row_shr:1
  KNN_DPP_STEP(0x112, 0xf)                                   // row_shr:2
  KNN_DPP_STEP(0x114, 0xf)                                   // row_shr:4
  KNN_DPP_STEP(0x118, 0xf)                                   // row_shr:8   -> lane 15 of every row holds the row's result
  KNN_DPP_STEP(0x142, 0xa)                                   // row_bcast:15 into rows 1, 3
  KNN_DPP_STEP(0x143, 0xc)                                   // row_bcast:31 into rows 2, 3 -> lane 63 holds the wave's result
#undef KNN_DPP_STEP
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// every lane offers one candidate (s, i); returns the wave's best in (ws, wi) (uniform); true on the lane that offered it (ids of real
// candidates are unique; all-sentinel waves return (-FLT_MAX, ID_NONE) and no winner)
__device__ __forceinline__ bool wave_best(float s, int i, float& ws, int& wi) {
  const unsigned key = f32_key(s);
  const unsigned kmax = wave_reduce_u32<true>(key);
  const unsigned imin = wave_reduce_u32<false>(key == kmax ? (unsigned)i : 0xffffffffu);
  wi = (int)imin;
  const bool won = key == kmax && i == wi && wi != ID_NONE;
  // the winner's ORIGINAL score bits (the key folds -0 onto +0: a -0.0 inner product must come out as -0.0, like oracle/flat_ip.c)
  const unsigned long long m = __ballot(won);
  ws = m ? __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(s), (int)__builtin_ctzll(m))) : key_f32(kmax);
  return won;
}

// One wave merges `nl` sorted lists resident in LDS (entry t of list l at s[l * stride + t], `len` entries each, (score desc, id asc))
// into the best `nout` of their union, in order: lane L owns lists L, L + 64, ... (LPL of them), keeps their heads in registers and
// re-reads only the winner's next entry per output — one LDS round trip per output instead of one per list.  out(o, score, id) is called
// by every lane with the same values.
template <int LPL, typename F>
__device__ __forceinline__ void wave_merge_lds(const float* s, const int* ix, int nl, int len, int stride, int nout, int lane, F&& out) {
  float hs[LPL]; int hi[LPL]; int ptr[LPL];
#pragma unroll
  for (int c = 0; c < LPL; ++c) {
    const int l = lane + 64 * c;
    ptr[c] = 0;
    hs[c] = -FLT_MAX; hi[c] = ID_NONE;
    if (l < nl && len > 0) { hs[c] = s[l * stride]; hi[c] = ix[l * stride]; }
  }
  for (int o = 0; o < nout; ++o) {
    float bs = hs[0]; int bi = hi[0]; int bc = 0;
#pragma unroll
    for (int c = 1; c < LPL; ++c)
      if (before(hs[c], hi[c], bs, bi)) { bs = hs[c]; bi = hi[c]; bc = c; }
    float ws; int wi;
    const bool won = wave_best(bs, bi, ws, wi);
    if (won) {
#pragma unroll
      for (int c = 0; c < LPL; ++c)
        if (c == bc) {
          const int l = lane + 64 * c;
          ++ptr[c];
          if (ptr[c] < len) { hs[c] = s[l * stride + ptr[c]]; hi[c] = ix[l * stride + ptr[c]]; }
          else { hs[c] = -FLT_MAX; hi[c] = ID_NONE; }
        }
    }
    out(o, ws, wi);
  }
}

// sorted-list insert, statically indexed (register resident).  Candidates arrive in ascending id
// order per lane, so a strict score compare keeps the lower id ahead on ties.
template <int KMAX>
__device__ __forceinline__ void topk_insert(float (&ls)[KMAX], int (&li)[KMAX], float s, int id) {
#pragma unroll
  for (int t = KMAX - 1; t >= 1; --t) {
    const bool gp = s > ls[t - 1];
    const bool g = s > ls[t];
    ls[t] = gp ? ls[t - 1] : (g ? s : ls[t]);
    li[t] = gp ? li[t - 1] : (g ? id : li[t]);
  }
  const bool g0 = s > ls[0];
  li[0] = g0 ? id : li[0];
  ls[0] = g0 ? s : ls[0];
}

struct KnnArgs {
  const void* q; int B;             // queries / index rows in the kernel's element type (fp32 exact, bf16 screening)
  const void* xb; int N; int D;
  int k;
  int nqt;                // query tiles
  int tiles_per_chunk;    // 128-row index tiles per chunk
  int nchunks;
  float* pdist; int* pidx;          // partial lists [nchunks][B][KMAX]   (nchunks > 1)
  float* dist; int64_t* idx;        // final [B][k]                        (nchunks == 1)
  // screening (knn_ip_topk_screened): candidate collection and the gated fallback
  const float* adist; const float* qnorm; float eps_scale;   // tau[q] = adist[q][k-1] - eps_scale * qnorm[q]
  int* cand; int* cnt; int cap;     // candidate ids [B][cap], counters [B]
  const int* run_flag;              // non-NULL: the launch is a no-op unless *run_flag != 0 (fallback after an overflow)
  int ring;                         // knn_stream_kernel: transpose stages per wave (1 | 2)
  int ctl_off;                      // knn_stream_kernel: LDS offset of the control words (block counter, shared score bounds)
  int pB, pq0;                      // knn_stream_kernel: partial lists addressed as [chunk][pB][KMAX] at query pq0 + q (0: [chunk][B][KMAX]; slices of a larger call)
  // k > 32: the result is produced 32 columns at a time.  Pass p writes columns [ocol, ocol + k) of the [B][ldo] outputs and
  // (AFTER) only ranks rows that come strictly AFTER the previous pass's last result (after_col) in the (score desc, id asc) order.
  int ldo, ocol, after_col;
  // knn_qs_kernel<.., POOL>: maxima of the approximate scores per 16-row block, M [4 * pairs][B], and per chunk, C [nchunks][B]
  float* pool_m; float* pool_c;
  int nsub;                         // sub-chunk maxima per chunk: min(QS_NSUB, pairs per chunk)
};

// E = float: exact scores (the product's definition).  E = __bf16: screening scores s^ from bf16-rounded operands
// (fp32 accumulation).  COLLECT: instead of keeping a top-k, append every row with s^ >= tau[q] to the query's
// candidate list — same MFMA path as the top-k pass, so s^ is bit-identical between the two passes.
template <int KMAX, typename E, bool COLLECT, bool AFTER = false>
__global__ __launch_bounds__(256, 2) void knn_partial_kernel(KnnArgs a) {
  if (a.run_flag != nullptr && *a.run_flag == 0) return;
  constexpr int MERGEB = 128 * 4 * KMAX * 8;
  constexpr int LDSB = GEMM_LDS > MERGEB ? GEMM_LDS : MERGEB;
  __shared__ __attribute__((aligned(16))) char smem[LDSB];
  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int w = wave_id(), wn = w >> 1, wm = w & 1;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = bid % a.nqt, chunk = bid / a.nqt;
  const int q0 = qt * 128;
  const int ntiles = (a.N + 127) / 128;
  const int t0 = chunk * a.tiles_per_chunk;
  const int t1 = min(t0 + a.tiles_per_chunk, ntiles);
  const int nks = a.D * (int)sizeof(E) / ROWB;           // K-stages of 128 bytes per row
  const int nstage = (t1 - t0) * nks;
  const E* Q = static_cast<const E*>(a.q);
  const E* X = static_cast<const E*>(a.xb);

  float ls[2][KMAX];
  int li[2][KMAX];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int t = 0; t < KMAX; ++t) { ls[j][t] = -FLT_MAX; li[j][t] = ID_NONE; }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 rx[4], rq[4];
  if (nstage > 0) {
    stage_load<E>(rx, X, a.D, t0 * 128, a.N, 0, tid);
    stage_load<E>(rq, Q, a.D, q0, a.B, 0, tid);
    stage_store<E>(rx, smem, tid);
    stage_store<E>(rq, smem + TILEB, tid);
  }
  __syncthreads();

  int tile = t0, ks = 0;
  for (int s = 0; s < nstage; ++s) {
    char* cur = smem + (s & 1) * STAGEB;
    char* nxt = smem + ((s & 1) ^ 1) * STAGEB;
    const bool more = (s + 1) < nstage;
    int ntile = tile, nksn = ks + 1;
    if (nksn == nks) { nksn = 0; ntile = tile + 1; }
    if (more) {
      stage_load<E>(rx, X, a.D, ntile * 128, a.N, nksn * ROWB, tid);
      stage_load<E>(rq, Q, a.D, q0, a.B, nksn * ROWB, tid);
    }
    stage_mma<E>(acc, cur, cur + TILEB, wn, wm, lane);
    if (ks == nks - 1) {
      // scores of index tile `tile` are complete: feed the per-lane top-k lists, reset.
      // Per query column j a lane holds 32 candidates (i, r) with ascending index id; a bitmask of
      // those beating the current k-th score is drained lowest-bit-first through ONE insertion site
      // (keeps everything statically indexed / register resident and the code small).
      const int nbase = tile * 128 + wn * 64 + 4 * half;
      if constexpr (COLLECT) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int qg = q0 + wm * 64 + j * 32 + r31;
          const bool qok = qg < a.B;
          const int qc = qok ? qg : a.B - 1;
          const float tau = a.adist[(int64_t)qc * a.k + (a.k - 1)] - a.eps_scale * a.qnorm[qc];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int n = nbase + i * 32 + (r & 3) + 8 * (r >> 2);
              if (qok && n < a.N && acc[i][j][r] >= tau) {
                const int pos = atomicAdd(a.cnt + qg, 1);
                if (pos < a.cap) a.cand[(int64_t)qg * a.cap + pos] = n;
              }
              acc[i][j][r] = 0.f;
            }
        }
      } else
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        uint32_t hits = 0;
        const float thr = ls[j][KMAX - 1];
        // quick reject (round 4): once the lists have warmed up almost no tile holds a candidate, and the per-score test below (row
        // bound, threshold, bit merge: PMC counted 12 VALU instructions per MFMA in the screening pass of configs[3]) is what the
        // kernel spent its time on.  The lane's 32 scores of this query column first go through a max tree (v_max3: half an
        // instruction per score); only a wave in which some lane's maximum beats its threshold looks at the scores one by one.
        float mxs = acc[0][j][0];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) mxs = fmaxf(mxs, acc[i][j][r]);
        if (__any(mxs > thr)) {
        float aS = 0.f; int aI = 0;
        if constexpr (AFTER) {                               // last result of the previous pass for this lane's query
          const int qg = q0 + wm * 64 + j * 32 + r31;
          const int qc = qg < a.B ? qg : a.B - 1;
          aS = a.dist[(int64_t)qc * a.ldo + a.after_col];
          const int64_t ai = a.idx[(int64_t)qc * a.ldo + a.after_col];
          aI = ai < 0 ? ID_NONE : (int)ai;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = nbase + i * 32 + (r & 3) + 8 * (r >> 2);
            bool ok = n < a.N && acc[i][j][r] > thr;
            if constexpr (AFTER) ok = ok && before(aS, aI, acc[i][j][r], n);
            hits |= ok ? (1u << (i * 16 + r)) : 0u;
          }
        if (__any(hits != 0)) {
          // wave-uniform walk over the 32 candidate slots (uniform index -> register-relative
          // addressing); lanes whose bit is set insert, ascending id order is preserved
          float cand[32];
#pragma unroll
          for (int r = 0; r < 16; ++r) { cand[r] = acc[0][j][r]; cand[16 + r] = acc[1][j][r]; }
#pragma unroll 1
          for (int bsel = 0; bsel < 32; ++bsel) {
            const bool mine = (hits >> bsel) & 1u;
            if (__any(mine)) {
              const int r = bsel & 15;
              const int n = nbase + (bsel >> 4) * 32 + (r & 3) + 8 * (r >> 2);
              const float sc = cand[bsel];
              if (mine) topk_insert<KMAX>(ls[j], li[j], sc, n);
            }
          }
        }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      }
    }
    if (more) {
      stage_store<E>(rx, nxt, tid);
      stage_store<E>(rq, nxt + TILEB, tid);
    }
    __syncthreads();
    tile = ntile; ks = nksn;
  }

  if constexpr (COLLECT) return;
  // ---- merge the 4 partial lists of every query through LDS (staging buffers are dead now)
  float* mS = reinterpret_cast<float*>(smem);                       // [128][4][KMAX]
  int* mI = reinterpret_cast<int*>(smem + 128 * 4 * KMAX * 4);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ql = wm * 64 + j * 32 + r31;
    const int src = wn * 2 + half;
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
      mS[(ql * 4 + src) * KMAX + t] = ls[j][t];
      mI[(ql * 4 + src) * KMAX + t] = li[j][t];
    }
  }
  __syncthreads();
  if (tid < 128) {
    const int qg = q0 + tid;
    if (qg < a.B) {
      const float* s0 = mS + tid * 4 * KMAX;
      const int* i0 = mI + tid * 4 * KMAX;
      int p0 = 0, p1 = 0, p2 = 0, p3 = 0;
      const int nout = (a.nchunks == 1) ? a.k : KMAX;
      for (int o = 0; o < nout; ++o) {
        float bs = -FLT_MAX; int bi = ID_NONE; int bsrc = -1;
        if (o < KMAX) {
          if (p0 < KMAX) { bs = s0[p0]; bi = i0[p0]; bsrc = 0; }
          if (p1 < KMAX && (bsrc < 0 || before(s0[KMAX + p1], i0[KMAX + p1], bs, bi))) { bs = s0[KMAX + p1]; bi = i0[KMAX + p1]; bsrc = 1; }
          if (p2 < KMAX && (bsrc < 0 || before(s0[2 * KMAX + p2], i0[2 * KMAX + p2], bs, bi))) { bs = s0[2 * KMAX + p2]; bi = i0[2 * KMAX + p2]; bsrc = 2; }
          if (p3 < KMAX && (bsrc < 0 || before(s0[3 * KMAX + p3], i0[3 * KMAX + p3], bs, bi))) { bs = s0[3 * KMAX + p3]; bi = i0[3 * KMAX + p3]; bsrc = 3; }
          p0 += (bsrc == 0); p1 += (bsrc == 1); p2 += (bsrc == 2); p3 += (bsrc == 3);
        }
        if (a.nchunks == 1) {
          a.dist[(int64_t)qg * a.ldo + a.ocol + o] = bs;
          a.idx[(int64_t)qg * a.ldo + a.ocol + o] = (bi == ID_NONE) ? (int64_t)-1 : (int64_t)bi;
        } else {
          const int64_t off = ((int64_t)chunk * a.B + qg) * KMAX + o;
          a.pdist[off] = bs;
          a.pidx[off] = bi;
        }
      }
    }
  }
}

// merge the per-chunk sorted lists of one query: one wave (= one workgroup) per query.  The query's lists — only their first
// min(k, KMAX) entries can reach the output — are staged into LDS with every load in flight at once, then merged by wave_merge_lds
// (lane L owns chunks L, L + 64, ...).  (Round 3's form re-read each head from global memory inside the output loop: one dependent
// round trip per output, 9-14 us per call.)
template <int KMAX>
__global__ __launch_bounds__(64) void knn_merge_kernel(const float* __restrict__ pdist, const int* __restrict__ pidx,
                                                       int B, int nchunks, int k, float* __restrict__ dist,
                                                       int64_t* __restrict__ idx, const int* __restrict__ run_flag, int ldo, int ocol) {
  if (run_flag != nullptr && *run_flag == 0) return;
  extern __shared__ __attribute__((aligned(16))) char msm[];
  constexpr int LPL = MAX_CHUNKS / 64;
  const int lane = threadIdx.x;
  const int qg = blockIdx.x;
  const int kk = k < KMAX ? k : KMAX;                    // entries of a list that can matter
  float* mS = reinterpret_cast<float*>(msm);             // [nchunks][kk]
  int* mI = reinterpret_cast<int*>(msm) + nchunks * kk;
  const int total = nchunks * kk;
  for (int e = lane; e < total; e += 64) {
    const int c = e / kk, t = e - c * kk;
    const int64_t off = ((int64_t)c * B + qg) * KMAX + t;
    mS[e] = pdist[off]; mI[e] = pidx[off];
  }
  __syncthreads();
  wave_merge_lds<LPL>(mS, mI, nchunks, kk, kk, k, lane, [&](int o, float ws, int wi) {
    if (lane == 0) {
      dist[(int64_t)qg * ldo + ocol + o] = (wi == ID_NONE) ? -FLT_MAX : ws;
      idx[(int64_t)qg * ldo + ocol + o] = (wi == ID_NONE) ? (int64_t)-1 : (int64_t)wi;
    }
  });
}
template <int KMAX>
void launch_knn_merge(const float* pdist, const int* pidx, int B, int nchunks, int k, float* dist, int64_t* idx, const int* run_flag,
                      int ldo, int ocol, hipStream_t s) {
  const int kk = k < KMAX ? k : KMAX;
  static_assert((size_t)MAX_CHUNKS * 32 * 8 <= 64 * 1024, "knn_merge: the staged lists must fit the default 64 KB dynamic LDS limit");
  hipLaunchKernelGGL((knn_merge_kernel<KMAX>), dim3((unsigned)B), dim3(64), (size_t)nchunks * kk * 8, s,
                     pdist, pidx, B, nchunks, k, dist, idx, run_flag, ldo, ocol);
}

// y = x / max(||x||_2, 1e-12) row-wise (F.normalize, infer_effocr.py:316); one wave per row
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, int64_t B, int D, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* xr = x + row * D;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) ss += xr[d] * xr[d];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f);
  for (int d = lane; d < D; d += 64) y[row * D + d] = xr[d] / nrm;
}

// dst[i] = src[rows[i]]  (IndexFlat.remove_ids compaction: gather of the kept rows)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ rows,
                                                          int64_t n, int D, float* __restrict__ dst) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = n * D;
  if (id >= total) return;
  const int64_t i = id / D;
  const int d = (int)(id - i * D);
  dst[id] = src[rows[i] * D + d];
}

int pick_kmax(int k) { return k <= 1 ? 1 : (k <= 16 ? 16 : 32); }   // k > 32: 32 columns per pass (knn_ip_topk)

struct Plan { int nqt, ntiles, tpc, nchunks, kmax; };

Plan make_plan(int64_t B, int64_t N, int k) {
  Plan p;
  p.kmax = pick_kmax(k);
  p.nqt = (int)((B + 127) / 128);
  p.ntiles = (int)((N + 127) / 128);
  if (p.ntiles < 1) p.ntiles = 1;
  // aim for ~2 resident workgroups per CU (512) without making chunks shorter than one tile
  int want = g_knn_wg_target / (p.nqt > 0 ? p.nqt : 1);
  if (want < 1) want = 1;
  if (want > MAX_CHUNKS) want = MAX_CHUNKS;
  if (want > p.ntiles) want = p.ntiles;
  p.tpc = (p.ntiles + want - 1) / want;
  p.nchunks = (p.ntiles + p.tpc - 1) / p.tpc;
  return p;
}

template <int KMAX, typename E>
int launch_knn(const KnnArgs& a, hipStream_t s) {
  if (a.after_col >= 0) hipLaunchKernelGGL((knn_partial_kernel<KMAX, E, false, true>), dim3((unsigned)(a.nqt * a.nchunks)), dim3(256), 0, s, a);
  else
  hipLaunchKernelGGL((knn_partial_kernel<KMAX, E, false>), dim3((unsigned)(a.nqt * a.nchunks)), dim3(256), 0, s, a);
  int rc = check_launch("knn_partial");
  if (rc != EFFOCR_OK || a.nchunks == 1) return rc;
  launch_knn_merge<KMAX>(a.pdist, a.pidx, a.B, a.nchunks, a.k, a.dist, a.idx, a.run_flag, a.ldo, a.ocol, s);
  return check_launch("knn_merge");
}
template <typename E>
int launch_knn_k(int kmax, const KnnArgs& a, hipStream_t s) {
  switch (kmax) {
    case 1: return launch_knn<1, E>(a, s);
    case 16: return launch_knn<16, E>(a, s);
    case 32: return launch_knn<32, E>(a, s);
  }
  return fail(EFFOCR_EINVAL, "knn: internal");
}

// ---- screening helpers --------------------------------------------------------------------------------------------
// queries -> bf16 copy + fp32 L2 norms; zero the candidate counters and the overflow flag.  One wave per query.
__global__ __launch_bounds__(256) void knn_prep_kernel(const float* __restrict__ q, int B, int D, __bf16* __restrict__ qb,
                                                       float* __restrict__ qnorm, int* __restrict__ cnt, int* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x == 0) *flag = 0;
  if (row >= B) return;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = q[(int64_t)row * D + d]; ss += v * v; qb[(int64_t)row * D + d] = (__bf16)v; }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
  if (lane == 0) { qnorm[row] = sqrtf(ss); cnt[row] = 0; }
}

__global__ __launch_bounds__(256) void convert_bf16_kernel(const float* __restrict__ src, int64_t n, __bf16* __restrict__ dst) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
    *reinterpret_cast<u32x2*>(dst + i) = pack4<__bf16>(v[0], v[1], v[2], v[3]);
  } else {
    for (int64_t j = i; j < n; ++j) dst[j] = (__bf16)src[j];
  }
}

// Candidates straight from pass 1's per-chunk lists (no second scan of the index).  Every chunk list holds the chunk's KMAX best
// approximate scores; a true top-k row r has s^_r >= tau = s^_(k) - 2 eps, so r is in its chunk's list unless KMAX rows of that chunk
// score at least s^_r >= tau — in which case the list's LAST entry is >= tau: that raises the overflow flag and the gated exact
// pass recomputes everything.  Otherwise the union of the list entries >= tau contains every true top-k row: same guarantee as the
// second scan, for the price of reading nchunks x KMAX x 8 bytes per query.  One wave per query, lanes over chunks.
template <int KMAX>
__global__ __launch_bounds__(256) void knn_collect_lists_kernel(const float* __restrict__ pdist, const int* __restrict__ pidx, int B, int nchunks,
                                                                int k, const float* __restrict__ adist, const float* __restrict__ qnorm,
                                                                float eps_scale, int* __restrict__ cand, int* __restrict__ cnt, int cap,
                                                                int* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int qg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qg >= B) return;
  const float tau = adist[(int64_t)qg * k + (k - 1)] - eps_scale * qnorm[qg];
  bool over = false;
  for (int c = lane; c < nchunks; c += 64) {
    const int64_t base = ((int64_t)c * B + qg) * KMAX;
    for (int t = 0; t < KMAX; ++t) {
      const float sc = pdist[base + t];
      const int id = pidx[base + t];
      if (id == ID_NONE || !(sc >= tau)) break;            // lists are sorted: nothing further qualifies
      const int pos = atomicAdd(cnt + qg, 1);
      if (pos < cap) cand[(int64_t)qg * cap + pos] = id;
      if (t == KMAX - 1) over = true;                     // the list is full of qualifying rows: the chunk may hold more
    }
  }
  if (__any(over) && lane == 0) atomicOr(flag, 1);
}

// Exact re-rank of one query's candidates: score = the ascending-k fp32 fmaf chain (the product's definition, what the
// fp32 MFMA kernel and oracle/flat_ip.c compute), order = (score desc, id asc).  One workgroup per query.
constexpr int RR_CAP = 512;
__global__ __launch_bounds__(256) void knn_rerank_kernel(const float* __restrict__ q, const float* __restrict__ xb, int D, int k,
                                                         const int* __restrict__ cand, const int* __restrict__ cnt, int cap,
                                                         float* __restrict__ dist, int64_t* __restrict__ idx, int* __restrict__ flag) {
  __shared__ float sS[RR_CAP];
  __shared__ int sI[RR_CAP];
  const int qg = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int n = cnt[qg];
  if (n > cap) { if (tid == 0) atomicOr(flag, 1); return; }   // overflow: the gated exact pass recomputes everything
  const float* qr = q + (int64_t)qg * D;
  for (int j = tid; j < RR_CAP; j += 256) {
    float sc = -FLT_MAX; int id = ID_NONE;
    if (j < n) {
      id = cand[(int64_t)qg * cap + j];
      const float* xr = xb + (int64_t)id * D;
      sc = 0.f;
      for (int d = 0; d < D; d += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(qr + d);
        const f32x4 b = *reinterpret_cast<const f32x4*>(xr + d);
        sc = fmaf(a[0], b[0], sc); sc = fmaf(a[1], b[1], sc); sc = fmaf(a[2], b[2], sc); sc = fmaf(a[3], b[3], sc);
      }
    }
    sS[j] = sc; sI[j] = id;
  }
  __syncthreads();
  if (tid >= 64) return;
  constexpr int PER = RR_CAP / 64;
  float ls[PER]; int li[PER];
#pragma unroll
  for (int t = 0; t < PER; ++t) { ls[t] = sS[lane + 64 * t]; li[t] = sI[lane + 64 * t]; }
  for (int o = 0; o < k; ++o) {
    float bs = -FLT_MAX; int bi = ID_NONE;
#pragma unroll
    for (int t = 0; t < PER; ++t)
      if (before(ls[t], li[t], bs, bi)) { bs = ls[t]; bi = li[t]; }
    float ws; int wi;
    wave_best(bs, bi, ws, wi);                             // (wave 0 is whole here: tid < 64)
    if (wi != ID_NONE) {
#pragma unroll
      for (int t = 0; t < PER; ++t)
        if (li[t] == wi) { ls[t] = -FLT_MAX; li[t] = ID_NONE; }   // ids are unique: exactly one owner
    }
    if (lane == 0) {
      dist[(int64_t)qg * k + o] = (wi == ID_NONE) ? -FLT_MAX : ws;
      idx[(int64_t)qg * k + o] = (wi == ID_NONE) ? (int64_t)-1 : (int64_t)wi;
    }
  }
}


// ---- small batches (B <= 32 queries: the reference's per-line calls, infer_effocr.py:313-317): the HBM-bound regime --------
// The 128-query tile kernel above spends 128 queries' worth of fp32 MFMA whatever B is (1.2 ms per pass over a 1M x 384
// index: 16 % of the HBM rate).  Here the queries are ONE 32-wide MFMA column tile that lives in LDS for the whole kernel,
// and the index streams from HBM straight into MFMA A-operand registers — no LDS staging, no barrier in the loop:
//   * the index is read in fully coalesced pieces (8 rows x 128 bytes per instruction; loading the row-per-lane MFMA layout
//     straight from memory was address-unit bound at 2.9 TB/s, and an LDS-DMA ring was latency bound at 3.5 TB/s: the bytes
//     in flight were capped by the LDS left next to the query image): four 32-row x 128-byte stages per wave wait in
//     REGISTERS (128 KB in flight per CU) and pass through a wave-private LDS buffer only to be transposed;
//   * v_mfma_f32_32x32x2_f32 takes A[row = lane & 31][k = lane >> 5]: the lane pair (r, r + 32) reads the SAME 16 bytes
//     X[row r][4m .. 4m+3] from the (XOR-swizzled) stage and feeds k = 4m + half, then 4m + 2 + half: two MFMAs per read, k
//     ascending, so a score is still bit for bit the ascending-k fmaf chain of oracle/flat_ip.c;
//   * queries: LDS image [m][half][query] of float2 (Q[q][4m + half], Q[q][4m + 2 + half]): one conflict-free ds_read_b64
//     per two MFMAs;
//   * a wave owns 32-row blocks of its workgroup's chunk round-robin; no workgroup barrier inside the loop;
//     2 x D/4 MFMAs per block = 16 B/clk/CU of index at the MFMA rate, i.e. the matrix pipe is ~2/3 busy at the HBM rate;
//   * per-lane sorted top-k lists as above; the 16 partial lists of a query (8 waves x 2 half-waves) merge through LDS,
//     chunks through knn_merge_kernel.
// NQT = 2 (33..64 queries; the ONNX driver's only call size is 64, infer_effocr_onnx_multi.py:157): two query column tiles share every
// index fragment (one LDS read, two MFMA pairs), twice the MFMA work per byte — 64 queries against 1M x 384 is MFMA-bound at 0.31 ms
// instead of 1.4 ms on the 128-query tile kernel.  The query images of both tiles (D * 256 bytes) leave room for ONE transpose
// stage per wave instead of two (LDS operations of a wave execute in order, so re-writing the stage behind its reads is safe).
#ifndef KNN_STREAM_NT
#define KNN_STREAM_NT 1                                  // the index rows pass once: non-temporal loads
#endif
constexpr int KS_THREADS = 512;
// -DKNN_STAMP (tools/ab_build.sh variant, never shipped): every workgroup of the streaming kernel records s_memtime at its milestones;
// tools/knn_timeline.py reads the last launch's table through effocr_debug_knn_stamps.
#ifdef KNN_STAMP
constexpr int KNN_STAMP_WGS = 256, KNN_STAMP_N = 8;
__device__ unsigned long long knn_stamps[KNN_STAMP_WGS * KNN_STAMP_N];
#define KNN_STAMP_AT(k) if (tid == 0) knn_stamps[(blockIdx.x & (KNN_STAMP_WGS - 1)) * KNN_STAMP_N + (k)] = __builtin_amdgcn_s_memtime();
#else
#define KNN_STAMP_AT(k)
#endif
// Q16 (<= 16 queries; NQT = 1): the query tile is 16 wide and the products run on v_mfma_f32_16x16x4_f32 — half the matrix time per index
// byte of the 32-wide tile, which at <= 32 queries costs as much as the HBM stream itself (16 B/clk/CU); the instruction adds its four k
// products in ascending k with one rounding each (tools/ubench/mfma16_order.hip: 256 of 256 results bit-identical to the fmaf chain), so
// the scores stay those of oracle/flat_ip.c.  A: lane (row l & 15, k l >> 4) reads ITS 4 bytes of the transposed stage; B: LDS image
// [m][k 0..3][16 queries]; C: lane holds query l & 15, rows 4 (l >> 4) + r of the 16-row group: four partial lists per query and wave.
// E = __bf16 (round 4): the SCREENING pass of knn_ip_topk_screened for 17..128 queries against a large index — the same stream over the
// bf16 copy of the index (half the bytes), v_mfma_f32_32x32x16_bf16 (a 32-row block x 32 queries x 64 k per stage and query tile: the matrix
// pipe idles), approximate scores s^ into the same lists.  Query image [qt][k16 step][k half][query] of 16-byte B-operand fragments; a stage
// is still 32 rows x 128 bytes (64 k), its A-operand fragments are the 16-byte chunks 2 ks + half of a row — one ds_read_b128 per k16 step.
// The shared bound is relaxed by the query's 2 eps (every row within 2 eps of the final k-th approximate score must survive into the lists:
// that is the candidate set of the exact re-rank), and chunk lists keep all KMAX entries (a full list of qualifying rows is the overflow signal).
template <int KMAX, int NQT, bool Q16 = false, typename E = float>
__global__ __launch_bounds__(KS_THREADS, 1) void knn_stream_kernel(KnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr bool BF = !std::is_same<E, float>::value;
  static_assert(!(BF && Q16), "the 16-query tile is an exact-search kernel");
  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int w = wave_id();
  const int chunk = blockIdx.x;
  const int D = a.D, nm = BF ? D / 8 : D / 4;                       // 16-byte pieces per query row
  const float* Q = static_cast<const float*>(a.q);
  const float* X = static_cast<const float*>(a.xb);
  f32x2* sQ = reinterpret_cast<f32x2*>(smem);                       // [NQT][nm][2][32]
  KNN_STAMP_AT(0)
  // Query image: every thread takes whole 16-byte pieces of the query rows (coalesced), FOUR loads in flight before the first LDS write
  // (round 3 read one float per thread and iteration: 12-24 dependent round trips in front of the first index byte).
  {
    constexpr int QW_ = Q16 ? 16 : 32 * NQT;                        // query slots of the image
    const int total = QW_ * nm;                                     // 16-byte pieces: (query slot, m)
    for (int base = 0; base < total; base += KS_THREADS * 4) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int id = base + u * KS_THREADS + tid;
        const int q = id / nm, m = id - q * nm;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (id < total && q < a.B) v[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.q) + ((int64_t)q * nm + m) * 16);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int id = base + u * KS_THREADS + tid;
        const int q = id / nm, m = id - q * nm;
        if (id < total) {
          if constexpr (Q16) {
            float* sQ1 = reinterpret_cast<float*>(smem);            // [D / 4][4][16]: element k = 4 m + kq of query q at (4 m + kq) * 16 + q
#pragma unroll
            for (int e = 0; e < 4; ++e) sQ1[(4 * m + e) * 16 + q] = v[u][e];
          } else if constexpr (BF) {
            const int qt = q >> 5, ql = q & 31;                     // piece m = 8 k: k16 step m / 2, k half m % 2 -> [qt][m][query] 16 B
            reinterpret_cast<f32x4*>(smem)[(qt * nm + m) * 32 + ql] = v[u];
          } else {
            const int qt = q >> 5, ql = q & 31;                     // [qt][m][half][query]: (Q[4m + h], Q[4m + 2 + h])
            sQ[(qt * nm + m) * 64 + ql] = f32x2{v[u][0], v[u][2]};
            sQ[(qt * nm + m) * 64 + 32 + ql] = f32x2{v[u][1], v[u][3]};
          }
        }
      }
    }
  }
  {
    unsigned* ctl = reinterpret_cast<unsigned*>(smem + a.ctl_off);
    if (tid < 4) ctl[tid] = 0u;                                     // [0]: block counter
    if (tid < (Q16 ? 16 : 32 * NQT)) ctl[4 + tid] = 0u;             // shared score bounds: key 0 = below every score
  }
  __syncthreads();
  KNN_STAMP_AT(1)

  float ls[NQT][KMAX];
  int li[NQT][KMAX];
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
    for (int t = 0; t < KMAX; ++t) { ls[qt][t] = -FLT_MAX; li[qt][t] = ID_NONE; }

  const int row_lo = chunk * a.tiles_per_chunk * 128;
  int row_hi = row_lo + a.tiles_per_chunk * 128;
  row_hi = row_hi < a.N ? row_hi : a.N;
  const f32x2* qp = sQ + half * 32 + r31;
  constexpr int NWAVE = KS_THREADS / 64;
  constexpr int WSTEP = NWAVE * 32;                                 // rows between a wave's consecutive blocks
  constexpr int STG = 4096;                                         // one stage: 32 rows x 128 bytes (32 k)
  constexpr int P = BF ? 3 : 4;                                     // stages in flight per wave, in REGISTERS (16 KB per wave, 128 KB per CU; bf16: 12 / 96 KB)
  const int nsl = BF ? D / 64 : D / 32;                             // stages (128-byte k slabs) per row block; P divides it (fp32: D % 128 == 0, bf16: D % 192 == 0)
  const int rowb = BF ? D * 2 : D * 4;                              // bytes per index row
  const int nbuf = a.ring;                                          // transpose stages per wave: 2, or 1 where the LDS has no room for two (launcher)
  char* stg = smem + (size_t)D * (BF ? 64 : 128) * NQT + (size_t)w * nbuf * STG;   // the wave's private transpose buffer
  // stream of this wave: stage t = (block t / nsl, slab t % nsl).  A stage is fetched by 4 fully coalesced 16-byte loads per
  // lane (lane -> row 8i + lane / 8, chunk lane % 8: 8 rows x 128 contiguous bytes per instruction), parked in registers
  // while P - 1 older stages are consumed, then transposed through the wave's LDS buffer into the row-per-lane MFMA layout.
  // The chunk position is XOR-swizzled with the row so that both the writes and the fragment reads spread over the banks.
  // Everything is wave-private: no workgroup barrier in the loop, and ordinary loads let the compiler count vmcnt itself.
  // Row blocks are handed out DYNAMICALLY inside the workgroup (round 4): an LDS counter, one atomic per 32-row block and wave.  With the
  // static round-robin of round 3 wave 0 sat 8 % of the kernel at the final barrier waiting for the other waves (s_memtime timeline,
  // tools/knn_timeline.py) — the waves of a CU do not get equal shares of the memory pipeline.  A wave's block numbers still ascend, so
  // its lanes still see their rows in ascending id order (the tie rule of topk_insert).
  const int nblk_wg = (row_hi - row_lo + 31) / 32;                   // 32-row blocks of this workgroup's chunk (<= 0: none)
  unsigned* sNext = reinterpret_cast<unsigned*>(smem + a.ctl_off);   // block counter (behind the stages; Q16: in the unused half of the image region)
  unsigned* sThr = sNext + 4;                                        // [QW * NQT] shared score bounds (f32_key), see below
  auto next_block = [&]() __attribute__((always_inline)) -> int {    // wave-uniform: row base of the wave's next block (>= row_hi: none)
    unsigned b = 0;
    if (lane == 0) b = __hip_atomic_fetch_add(sNext, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    b = (unsigned)__builtin_amdgcn_readfirstlane((int)b);
    return (int)b < nblk_wg ? row_lo + (int)b * 32 : INT_MAX - 64;
  };
  f32x4 rg[P][4];
  auto fetch = [&](f32x4 (&r)[4], int r0i, int sl) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = r0i + 8 * i + (lane >> 3);
      row = row < a.N ? row : a.N - 1;                              // clamp: rows past the end are masked below
#if KNN_STREAM_NT
      r[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(X) + (int64_t)row * rowb + sl * 128 + ((lane & 7) << 4)));
#else
      r[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(X) + (int64_t)row * rowb + sl * 128 + ((lane & 7) << 4));
#endif
    }
  };
  int frow = next_block(), fs = 0;                                  // (row base, slab) of the next stage to fetch
  int r0 = frow;                                                    // row base of the block being consumed
  // blocks acquired by the fetch side and not yet begun by the consuming side (the fetch runs P stages = up to one whole block ahead): a
  // two-entry queue of row bases; the end-of-chunk sentinel is queued once
  int q0 = 0, q1 = 0, pend = 0;
  auto fetch_wrap = [&]() __attribute__((always_inline)) {
    fs = 0;
    if (frow < row_hi) {
      frow = next_block();
      if (pend == 0) q0 = frow; else q1 = frow;
      ++pend;
    }
  };
#pragma unroll
  for (int u = 0; u < P; ++u) {
    if (frow < row_hi) fetch(rg[u], frow, fs);
    if (++fs == nsl) fetch_wrap();
  }
  f32x16 acc[NQT];
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[qt][r] = 0.f;
  f32x4 acc16[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};    // Q16: row groups 0-15 / 16-31 of the block
  const int q16 = lane & 15, kq = lane >> 4;
  const float* qp16 = reinterpret_cast<const float*>(smem) + kq * 16 + q16;
  int sl = 0;
  const int wr = lane >> 3, wc = lane & 7;
  // Shared score bounds (round 4).  A lane's list only sees 1/16 .. 1/32 of the workgroup's rows, so its own KMAX-th score is a loose
  // filter: at 64 queries about every second (block, slot) ran an insertion.  Any list's KMAX-th entry is a lower bound of the final k-th
  // score of that query (>= KMAX >= k rows score at least that), so the lists of a query publish theirs with an LDS atomic max
  // (order-preserving keys) once per block and filter with the workgroup's best bound: a row below it cannot reach the output; rows
  // EQUAL to it still can (ties rank by id) and pass.  The lists are then no longer each chunk part's complete top-KMAX, but their union
  // still holds the chunk's top-k, which is all the merges need.
  float sthr[NQT], eps2[NQT];                                       // eps2: screening only — the band below the bound that must survive (2 eps of the lane's query)
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt) {
    sthr[qt] = -FLT_MAX; eps2[qt] = 0.f;
    if constexpr (BF) { const int q = qt * 32 + r31; eps2[qt] = a.eps_scale * a.qnorm[q < a.B ? q : a.B - 1]; }
  }
  while (r0 < row_hi) {
#pragma unroll
    for (int u = 0; u < P; ++u) {
      char* buf = stg + (u & (nbuf - 1)) * STG;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = 8 * i + wr;
        *reinterpret_cast<f32x4*>(buf + rl * 128 + ((wc ^ (rl & 7)) << 4)) = rg[u][i];
      }
      if (frow < row_hi) fetch(rg[u], frow, fs);                     // the slot's next stage (P stages ahead)
      if (++fs == nsl) fetch_wrap();
      if constexpr (Q16) {
        float xa[2][8];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const int row = 16 * g + q16;
            xa[g][m] = *reinterpret_cast<const float*>(buf + row * 128 + ((m ^ (row & 7)) << 4) + kq * 4);
          }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const float qv = qp16[((sl + u) * 8 + m) * 64];
          acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0][m], qv, acc16[0], 0, 0, 0);
          acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[1][m], qv, acc16[1], 0, 0, 0);
        }
      } else if constexpr (BF) {
        typedef typename Op16<__bf16>::V8 V8;
        V8 xa[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xa[ks] = *reinterpret_cast<const V8*>(buf + r31 * 128 + (((2 * ks + half) ^ (r31 & 7)) << 4));
        const V8* qb16 = reinterpret_cast<const V8*>(smem) + half * 32 + r31;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
          for (int qt = 0; qt < NQT; ++qt) {
            const V8 qv = qb16[(qt * nm + 2 * ((sl + u) * 4 + ks)) * 32];
            acc[qt] = Op16<__bf16>::mfma(xa[ks], qv, acc[qt]);
          }
        }
      } else {
      f32x4 xv[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) xv[m] = *reinterpret_cast<const f32x4*>(buf + r31 * 128 + ((m ^ (r31 & 7)) << 4));
#pragma unroll
      for (int m = 0; m < 8; ++m) {
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
          const f32x2 qv = qp[(qt * nm + (sl + u) * 8 + m) * 64];
          acc[qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? xv[m][1] : xv[m][0], qv[0], acc[qt], 0, 0, 0);
          acc[qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? xv[m][3] : xv[m][2], qv[1], acc[qt], 0, 0, 0);
        }
      }
      }
    }
    sl += P;
    if (sl == nsl) {
      if constexpr (Q16) {
        // C layout: col = query (lane & 15), rows 16 g + 4 (lane >> 4) + r, ascending with (g, r)
        uint32_t hits = 0;
        const float thr = ls[0][KMAX - 1], thg = sthr[0];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = r0 + 16 * g + 4 * kq + r;
            hits |= (n < a.N && acc16[g][r] > thr && acc16[g][r] >= thg) ? (1u << (4 * g + r)) : 0u;
          }
        if (__any(hits != 0)) {
#pragma unroll
          for (int bsel = 0; bsel < 8; ++bsel) {
            const bool mine = (hits >> bsel) & 1u;
            if (__any(mine)) {
              const int n = r0 + 16 * (bsel >> 2) + 4 * kq + (bsel & 3);
              const float sc = acc16[bsel >> 2][bsel & 3];
              if (mine) topk_insert<KMAX>(ls[0], li[0], sc, n);
            }
          }
        }
        acc16[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc16[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
          const unsigned own = f32_key(ls[0][KMAX - 1]);
          const unsigned old = __hip_atomic_fetch_max(sThr + q16, own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          sthr[0] = key_f32(old > own ? old : own);
        }
      } else {
      // C layout: col = query (r31), rows = index rows (r & 3) + 8 (r >> 2) + 4 half, ascending with r
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) {
        uint32_t hits = 0;
        const float thr = ls[qt][KMAX - 1], thg = sthr[qt] - eps2[qt];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = r0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          hits |= (n < a.N && acc[qt][r] > thr && acc[qt][r] >= thg) ? (1u << r) : 0u;
        }
        if (__any(hits != 0)) {
          float cand[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) cand[r] = acc[qt][r];
#pragma unroll 1
          for (int bsel = 0; bsel < 16; ++bsel) {
            const bool mine = (hits >> bsel) & 1u;
            if (__any(mine)) {
              const int n = r0 + (bsel & 3) + 8 * (bsel >> 2) + 4 * half;
              const float sc = cand[bsel];
              if (mine) topk_insert<KMAX>(ls[qt], li[qt], sc, n);
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[qt][r] = 0.f;
        {
          const unsigned own = f32_key(ls[qt][KMAX - 1]);
          const unsigned old = __hip_atomic_fetch_max(sThr + qt * 32 + r31, own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          sthr[qt] = key_f32(old > own ? old : own);
        }
      }
      }
      sl = 0; r0 = q0; q0 = q1; --pend;                               // (the fetch side is ahead: the next block, or the sentinel, is queued)
    }
  }

  // ---- merge the partial lists of every query through LDS (the query image is dead): the lists go to LDS as before, then ONE WAVE per
  // query merges them with wave_merge_lds (lane = list) — round 3 had one THREAD per query walk 16-32 lists with a dependent LDS
  // round trip per list and output (~20 us of a 0.3 ms search).  Chunks keep min(k, KMAX) entries: what the chunk merge can use.
  KNN_STAMP_AT(2)
  __syncthreads();
  KNN_STAMP_AT(3)
  constexpr int NSRC = (KS_THREADS / 64) * (Q16 ? 4 : 2);
  constexpr int QW = Q16 ? 16 : 32;                                 // queries per tile
  // odd strides: list stride KMAX + 1, query stride NSRC (KMAX + 1) + 1 words — the lanes of a wave write entry t of 64 different lists
  // and wave_merge_lds reads the heads of 16-32 lists at once; with power-of-two strides both were 16-way bank conflicts (4.7 us)
  constexpr int LS = KMAX + 1, QS = NSRC * LS + 1;
  float* mS = reinterpret_cast<float*>(smem);                       // [QW * NQT][NSRC][KMAX] with those strides
  int* mI = reinterpret_cast<int*>(smem) + QW * NQT * QS;
  if constexpr (Q16) {
    const int src = w * 4 + kq;
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
      mS[q16 * QS + src * LS + t] = ls[0][t];
      mI[q16 * QS + src * LS + t] = li[0][t];
    }
  } else {
    const int src = w * 2 + half;
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
      for (int t = 0; t < KMAX; ++t) {
        mS[(qt * 32 + r31) * QS + src * LS + t] = ls[qt][t];
        mI[(qt * 32 + r31) * QS + src * LS + t] = li[qt][t];
      }
  }
  __syncthreads();
  KNN_STAMP_AT(4)
  const int kk = BF ? KMAX : (a.k < KMAX ? a.k : KMAX);              // (screening: collect_lists reads whole lists)
  const int nout = (a.nchunks == 1) ? a.k : kk;
  const int nq = a.B < QW * NQT ? a.B : QW * NQT;
  for (int q = w; q < nq; q += NWAVE) {                              // wave-uniform: whole waves enter wave_merge_lds
    wave_merge_lds<1>(mS + q * QS, mI + q * QS, NSRC, KMAX, LS, nout, lane, [&](int o, float ws, int wi) {
      if (lane == 0) {
        if (a.nchunks == 1) {
          a.dist[(int64_t)q * a.ldo + a.ocol + o] = (wi == ID_NONE) ? -FLT_MAX : ws;
          a.idx[(int64_t)q * a.ldo + a.ocol + o] = (wi == ID_NONE) ? (int64_t)-1 : (int64_t)wi;
        } else {
          const int64_t off = ((int64_t)chunk * (a.pB ? a.pB : a.B) + a.pq0 + q) * KMAX + o;
          a.pdist[off] = (wi == ID_NONE) ? -FLT_MAX : ws;
          a.pidx[off] = wi;
        }
      }
    });
  }
  KNN_STAMP_AT(5)
}

template <int KMAX, int NQT, bool Q16 = false, typename E = float>
int launch_knn_stream(const KnnArgs& a_in, hipStream_t s, bool merge = true) {
  KnnArgs a = a_in;
  constexpr int NSRC = (KS_THREADS / 64) * (Q16 ? 4 : 2), QW = Q16 ? 16 : 32;
  const size_t q_bytes = (size_t)a.D * (std::is_same<E, float>::value ? 128 : 64) * NQT, m_bytes = (size_t)QW * NQT * (NSRC * (KMAX + 1) + 1) * 8;
  // control words (block counter + shared bounds): the 16-query image fills only half of its region — they live in the other half;
  // otherwise behind the stages.  Two transpose stages per wave where they fit next to one query tile's image, else one (LDS operations
  // of a wave execute in order, so re-writing a stage behind its reads is safe): 1M x 768 at 17..32 queries, and every two-tile launch.
  const size_t c_bytes = Q16 ? 0 : 16 + (size_t)QW * NQT * 4;
  int nbuf = NQT == 1 ? 2 : 1;
  if (q_bytes + (size_t)(KS_THREADS / 64) * nbuf * 4096 + c_bytes > 160 * 1024) nbuf = 1;
  const size_t s_bytes = q_bytes + (size_t)(KS_THREADS / 64) * nbuf * 4096 + c_bytes;
  a.ring = nbuf;
  a.ctl_off = Q16 ? a.D * 64 : (int)(s_bytes - c_bytes);
  if (s_bytes > 160 * 1024 || m_bytes > 160 * 1024) return fail(EFFOCR_EUNSUPPORTED, "knn(stream): embedding dim / k too large for the LDS image");
  const size_t lds = s_bytes > m_bytes ? s_bytes : m_bytes;
  // per launch: the attribute belongs to the (function, device) pair and a process may search on several GPUs; the call is cheap
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_stream_kernel<KMAX, NQT, Q16, E>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return fail(EFFOCR_EHIP, "knn(stream): hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
  hipLaunchKernelGGL((knn_stream_kernel<KMAX, NQT, Q16, E>), dim3((unsigned)a.nchunks), dim3(KS_THREADS), lds, s, a);
  int rc = check_launch("knn_stream");
  if (rc != EFFOCR_OK || a.nchunks == 1 || !merge) return rc;
  launch_knn_merge<KMAX>(a.pdist, a.pidx, a.B, a.nchunks, a.k, a.dist, a.idx, a.run_flag, a.ldo, a.ocol, s);
  return check_launch("knn_merge");
}
int launch_knn_stream_k(int kmax, int nqt, const KnnArgs& a, hipStream_t s) {
  if (nqt == 2) {
    switch (kmax) {
      case 1: return launch_knn_stream<1, 2>(a, s);
      case 16: return launch_knn_stream<16, 2>(a, s);
    }
    return fail(EFFOCR_EINVAL, "knn: internal");
  }
  if (a.B <= 16 && g_knn_q16) {
    switch (kmax) {
      case 1: return launch_knn_stream<1, 1, true>(a, s);
      case 16: return launch_knn_stream<16, 1, true>(a, s);
      case 32: return launch_knn_stream<32, 1, true>(a, s);
    }
    return fail(EFFOCR_EINVAL, "knn: internal");
  }
  switch (kmax) {
    case 1: return launch_knn_stream<1, 1>(a, s);
    case 16: return launch_knn_stream<16, 1>(a, s);
    case 32: return launch_knn_stream<32, 1>(a, s);
  }
  return fail(EFFOCR_EINVAL, "knn: internal");
}
// ---- Q-stationary bf16 screening kernel (round 5) ---------------------------------------------------------------------------------
// Pass 1 of knn_ip_topk_screened for ANY index size when the caller keeps a FRAGMENT-BLOCKED bf16 copy of the index
// (effocr_convert_bf16_blocked: cells [row / 32][k chunk of 8][row % 32][16 B], the layout of the encoder's weight copies).  The
// 128-query tile kernel (knn_partial_kernel<bf16>) restages BOTH operands through LDS every 64 k with a workgroup barrier per stage
// and ran at 22 % of the bf16 MFMA peak on BASELINE configs[3] (1M x 768, 1024 queries: 2.9 ms) and at 95 TFLOP/s on configs[1]'s own
// search (10 k x 384: 83 us of a 141 us, seven-launch chain for 7.9 GFLOP).  Here the geometry is the projection phase of qkvattn.hip:
//   * the QUERIES are stationary: wave w holds QT tiles of 32 queries as MFMA B-operand fragments in registers for the whole kernel
//     (D / 16 fragments per tile: 96 registers at D = 384 with QT = 2 tiles = 256 queries per workgroup, 192 at D = 768 with QT = 1);
//   * the INDEX streams: stage = 64 rows x 128 k (16 KB) through a 6-slot LDS ring by LDS-DMA, five stages ahead, counted vmcnt + one
//     raw barrier in the middle of a stage; the blocked copy makes a stage a verbatim copy of 512-byte cells, so the A-operand
//     fragment reads (row r31, k half) are conflict-free ds_read_b128 — no transposes, no swizzles, no bank conflicts;
//   * swapped MFMA (rows = index rows, columns = queries): a lane's 16 accumulators of a tile belong to ONE query and 16 rows.
// The screened search stays BIT-IDENTICAL to the exact one (same eps bound: bf16 operand rounding + fp32 accumulation, any summation order).
constexpr int QS_STAGE = 16384, QS_RING = 6;
constexpr int QS_NSUB = 8;            // sub-chunk maxima per chunk (pass 1 -> knn_pool_bound_kernel)
// NO lists at all.  Lane-local lists cost more than the products they rank — the chip holds 65 536 lanes, so
// every query owns 64 lists of N / 64 rows each and a list takes KMAX (1 + ln(N / 64 / KMAX)) insertions of ~70 instructions that no
// other lane of the wave shares: measured 0.86 ms of the 2.46 ms kernel at 1M x 768 x 1024 queries, and 80 us of a 131 us search at
// 10 000 rows (where a third of all scores gets inserted).  Instead a lane writes the MAXIMUM of its 16 accumulators per (row tile,
// query): one value per 16-row block, M[block][query] (15 v_max + one coalesced store per tile), and keeps a running maximum per
// chunk, C[chunk][query].  Selection happens afterwards on 1/16 of the values and by THRESHOLD, not by insertion (knn_pool_collect_kernel,
// knn_pool_rerank_kernel below).
// (The list-keeping form of this kernel — round 5's first version: per-lane lists with a per-lane hit walk through an LDS scratch, merged
// per chunk and fed to the merge / collect / re-rank chain — measured 2.46 ms / 131 us at those two shapes and was removed.)
// FINE: maxima per 4 consecutive rows (a register quad) instead of per 16 — four times the block maxima (N * B bytes), a quarter of the
// rows the re-rank has to re-score per surviving block; used where that array stays small (N * B <= 64 M: BASELINE configs[1]'s own search).
template <int D, int QT, bool FINE = false>
__global__ __launch_bounds__(256, 1) void knn_qs_kernel(KnnArgs a) {
  constexpr bool POOL = true;
  typedef bf16x8 V8;
  constexpr int KC = D / 8, KT = D / 128, NXF = D / 16, R = QS_RING;
  static_assert(D % 128 == 0, "knn_qs: D must be a multiple of 128");
  __shared__ __attribute__((aligned(16))) char smem[R * QS_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int w = wave_id();
  // logical id = chunk * nqt + query group: the query groups of a chunk are consecutive logical ids = the same XCD (xcd_remap) and
  // dispatched together, so a chunk's index rows leave HBM once and serve every query group out of that XCD's L2
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int qg = lid % a.nqt, chunk = lid / a.nqt;
  const int npairs = (a.N + 63) / 64;
  const int p0 = chunk * a.tiles_per_chunk;                          // 64-row pairs [p0, p1) of this chunk
  const int p1 = min(p0 + a.tiles_per_chunk, npairs);
  const int nst = (p1 - p0) * KT;                                    // ring stages of this workgroup (>= KT)
  const int sublen = (a.tiles_per_chunk + a.nsub - 1) / a.nsub;    // pairs per sub-chunk (the maxima the bound L(q) is taken over)
  const char* Xb = static_cast<const char*>(a.xb);
  const float* Q = static_cast<const float*>(a.q);                    // fp32 queries, rounded to bf16 here (no separate conversion launch)

  // ---- query fragments: lane (query r31 of tile, k half) holds k = 16 t + 8 half .. + 7 for every k16 step t
  V8 qf[QT][NXF];
  const int q0 = (qg * 4 + w) * QT * 32;                             // first query of this wave
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qi = q0 + qt * 32 + r31;
    const float* qr = Q + (int64_t)(qi < a.B ? qi : a.B - 1) * D + 8 * half;
#pragma unroll
    for (int t = 0; t < NXF; ++t) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(qr + 16 * t), hi = *reinterpret_cast<const f32x4*>(qr + 16 * t + 4);
      const u32x2 p0 = pack4<__bf16>(lo[0], lo[1], lo[2], lo[3]), p1 = pack4<__bf16>(hi[0], hi[1], hi[2], hi[3]);
      qf[qt][t] = __builtin_bit_cast(V8, u32x4{p0[0], p0[1], p1[0], p1[1]});
    }
  }

  // ---- ring: stage s = (pair p0 + s / KT, k slice s % KT); wave w copies row block w >> 1, k chunks (w & 1) * 8 .. + 8: 4 pieces of 1 KB
  int ip = p0, ikt = 0, islot = 0, issued = 0;
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned sW_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto issue_stage = [&]() __attribute__((always_inline)) {
    const char* src = Xb + ((size_t)(2 * ip + (w >> 1)) * KC + ikt * 16 + (w & 1) * 8) * 512 + lane16;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(sW_lds + (unsigned)(islot * QS_STAGE) + (unsigned)(((w >> 1) * 16 + (w & 1) * 8) * 512)));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    islot = islot + 1 == R ? 0 : islot + 1;
    if (++ikt == KT) { ikt = 0; ++ip; }
    ++issued;
  };
  for (int s0 = 0; s0 < R - 1 && s0 < nst; ++s0) issue_stage();

  float cmax[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) cmax[qt] = -FLT_MAX;

  // stage 0 has landed (own pieces: everything issued behind it may stay in flight; fewer than R - 1 stages in all: wait for all)
  if (nst >= R - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 4) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  int slot = 0, s = 0;
  const int fo = half * 512 + r31 * 16;                              // the lane's fragment inside a cell pair
  // A-operand fragments are requested TWO k16 steps ahead of their MFMAs: with one wave per SIMD nothing else hides the LDS latency,
  // and at QT = 1 a k16 step is only two MFMAs (64 cycles) long while the four waves' reads run the LDS at its full rate
  V8 f0 = *reinterpret_cast<const V8*>(smem + fo), f1 = *reinterpret_cast<const V8*>(smem + fo + 16 * 512);
  V8 g0 = *reinterpret_cast<const V8*>(smem + fo + 2 * 512), g1 = *reinterpret_cast<const V8*>(smem + fo + 18 * 512);
#pragma unroll 1
  for (int p = p0; p < p1; ++p) {
    f32x16 acc[2][QT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][qt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const char* st = smem + slot * QS_STAGE + fo;
      const int nslot = slot + 1 == R ? 0 : slot + 1;
      const char* stn = smem + nslot * QS_STAGE + fo;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks == 4) {
          // middle of stage s: stage s + 1 has landed (own pieces), and past the barrier everybody's; every wave is done with stage
          // s - 1, whose slot takes stage s + R - 1.  In the steady state R - 3 younger stages may stay in flight; at the end of
          // the stream (nothing more to issue) wait for everything.
          if (issued - (s + 2) >= R - 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * 4) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (issued < nst) issue_stage();
        }
        V8 h0, h1;
        if (ks < 6) {
          h0 = *reinterpret_cast<const V8*>(st + (2 * (ks + 2)) * 512);
          h1 = *reinterpret_cast<const V8*>(st + (16 + 2 * (ks + 2)) * 512);
        } else {                                           // stage s + 1 landed for everybody at this stage's barrier (past the end: a dead read)
          h0 = *reinterpret_cast<const V8*>(stn + (2 * (ks - 6)) * 512);
          h1 = *reinterpret_cast<const V8*>(stn + (16 + 2 * (ks - 6)) * 512);
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          acc[0][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, qf[qt][kt * 8 + ks], acc[0][qt], 0, 0, 0);
          acc[1][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, qf[qt][kt * 8 + ks], acc[1][qt], 0, 0, 0);
        }
        f0 = g0; f1 = g1; g0 = h0; g1 = h1;
      }
      slot = nslot; ++s;
    }
    if constexpr (POOL) {
      // block b = 4 p + 2 i + half = the 16 rows 64 p + 32 i + 4 half + (r & 3) + 8 (r >> 2) this lane holds of row tile i
      const bool lastp = p == npairs - 1;                              // only the last pair can hold rows >= N (zero rows of the blocked copy)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const int qi = q0 + qt * 32 + r31;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if constexpr (FINE) {
            // block b4 = (4 p + 2 i + half) * 4 + g = the 4 rows 64 p + 32 i + 4 half + 8 g + 0..3 (registers 4g .. 4g + 3)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float m = -FLT_MAX;
#pragma unroll
              for (int e = 0; e < 4; ++e)
                m = (!lastp || p * 64 + i * 32 + 4 * half + 8 * g + e < a.N) ? fmaxf(m, acc[i][qt][4 * g + e]) : m;
              if (qi < a.B) a.pool_m[(int64_t)((p * 4 + i * 2 + half) * 4 + g) * a.B + qi] = m;
              cmax[qt] = fmaxf(cmax[qt], m);
            }
          } else {
          float m = -FLT_MAX;
          if (lastp) {
#pragma unroll
            for (int r = 0; r < 16; ++r) m = (p * 64 + i * 32 + 4 * half + (r & 3) + 8 * (r >> 2) < a.N) ? fmaxf(m, acc[i][qt][r]) : m;
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[i][qt][r]);
          }
          if (qi < a.B) a.pool_m[(int64_t)(p * 4 + i * 2 + half) * a.B + qi] = m;
          cmax[qt] = fmaxf(cmax[qt], m);
          }
        }
      }
      // end of a sub-chunk (QS_NSUB per chunk): its maximum per query -> C[chunk * QS_NSUB + sub][q], running maximum reset
      if ((p - p0 + 1) % sublen == 0 || p == p1 - 1) {
        const int sub = (p - p0) / sublen;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const float m = fmaxf(cmax[qt], __shfl_xor(cmax[qt], 32, 64));
          const int qi = q0 + qt * 32 + r31;
          if (half == 0 && qi < a.B) a.pool_c[(int64_t)qi * (a.nchunks * a.nsub) + chunk * a.nsub + sub] = m;   // [query][sub-chunk]: one coalesced row per query for the bound kernel
          cmax[qt] = -FLT_MAX;
        }
      }
    }
  }
  // sub-chunk slots this chunk did not reach (a short last chunk): no block there
  for (int sub = (p1 - p0 + sublen - 1) / sublen; sub < a.nsub; ++sub)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int qi = q0 + qt * 32 + r31;
      if (half == 0 && qi < a.B) a.pool_c[(int64_t)qi * (a.nchunks * a.nsub) + chunk * a.nsub + sub] = -FLT_MAX;
    }
}

// fp32 [N][D] row-major -> bf16 fragment-blocked [ceil(N / 64) * 2][D / 8][32][8]; rows >= N are zero.  One thread per 16-byte cell slot.
__global__ __launch_bounds__(256) void convert_bf16_blocked_kernel(const float* __restrict__ src, int64_t N, int D, __bf16* __restrict__ dst) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int kc = D / 8;
  const int64_t nrb = (N + 63) / 64 * 2;
  if (id >= nrb * kc * 32) return;
  const int row = (int)(id & 31);
  const int64_t cell = id >> 5;
  const int c = (int)(cell % kc);
  const int64_t n = (cell / kc) * 32 + row;
  u32x4 o = {0u, 0u, 0u, 0u};
  if (n < N) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + n * D + c * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + n * D + c * 8 + 4);
    const u32x2 lo = pack4<__bf16>(a[0], a[1], a[2], a[3]), hi = pack4<__bf16>(b[0], b[1], b[2], b[3]);
    o = u32x4{lo[0], lo[1], hi[0], hi[1]};
  }
  *reinterpret_cast<u32x4*>(dst + id * 8) = o;
}

// ---- selection over the block maxima (pass 1 = knn_qs_kernel<.., POOL>) ------------------------------------------------------------
// L(q) = the k-th largest CHUNK maximum of query q is a lower bound of m_k(q), the k-th largest BLOCK maximum (chunk maxima are block
// maxima of distinct blocks), and m_k <= s^_(k): the k blocks with the largest maxima hold k distinct rows.  Hence a true top-k row r
// (s_r >= s_(k), |s^ - s| <= eps) has s^_r >= s_(k) - eps >= s^_(k) - 2 eps >= m_k - 2 eps >= L - 2 eps, and so has its block's maximum.
//   collect  every block with M[b][q] >= L(q) - 2 eps(q) -> the query's entry list (value, block), <= POOL_CAP entries (else: overflow
//            flag -> the gated exact pass).  Workgroup = 64 queries (lane = query: coalesced rows of M) x a range of blocks.
//   rerank   per query: m_k = the k-th largest collected value (all blocks >= L - 2 eps are there, so the k largest are), survivors =
//            entries >= m_k - 2 eps, exact ascending-k fmaf chains of the survivors' 16 rows each, (score desc, id asc) top k.
// Results stay bit-identical to the exact search: the candidate ROWS are a superset of the true top k, their scores the product's own.
constexpr int POOL_CAP = 512;          // collected blocks per query
constexpr int POOL_ROWS = 2048;        // rows re-scored per query in stage A (128 surviving blocks; the 2 eps band of a 768-wide index holds
                                       // 35 rows on average and 58 at most over 256 queries against 1M random rows — each in its own block)
// thr[q] = L(q) - 2 eps(q), L = the k-th largest of the nsub sub-chunk maxima of the query.  One workgroup per query, by RANK COUNTING:
// a thread holds one value (nsub <= 256: BASELINE configs[1] and [3] alike; more: several) and counts the values that rank before it —
// one memory round trip and a 256-step LDS loop (a per-lane sorted list over 64 queries per workgroup took 24 us on 16 workgroups).
__global__ __launch_bounds__(256) void knn_pool_bound_kernel(const float* __restrict__ C, int B, int nsub, int k, const float* __restrict__ qv, int D,
                                                             float* __restrict__ qnorm, float eps_scale, float* __restrict__ thr,
                                                             int* __restrict__ cnt, int* __restrict__ flag) {
  __shared__ float sC[MAX_CHUNKS * QS_NSUB];
  __shared__ float sL, sQ2[4];
  const int q = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) { sL = -FLT_MAX; cnt[q] = 0; if (q == 0) *flag = 0; }   // (this kernel also does what knn_prep did for the older chains:
  for (int e = tid; e < nsub; e += 256) sC[e] = C[(int64_t)q * nsub + e];   //  candidate counters, overflow flag, the query's L2 norm)
  float ss = 0.f;
  for (int d = tid; d < D; d += 256) { const float v = qv[(int64_t)q * D + d]; ss += v * v; }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
  if ((tid & 63) == 0) sQ2[tid >> 6] = ss;
  __syncthreads();
  const float qn = sqrtf((sQ2[0] + sQ2[1]) + (sQ2[2] + sQ2[3]));
  const int kk = k < nsub ? k : nsub;
  for (int e = tid; e < nsub; e += 256) {
    const float v = sC[e];
    int rank = 0;
    for (int j = 0; j < nsub; ++j) { const float u = sC[j]; rank += (u > v || (u == v && j < e)) ? 1 : 0; }
    if (rank == kk - 1 && v == v) sL = v;                  // exactly one entry of a NaN-free set has this rank
  }
  __syncthreads();
  if (tid == 0) { thr[q] = sL - eps_scale * qn; qnorm[q] = qn; }   // written on every call: never a stale threshold
}

// Workgroup = 64 queries (lane = query: coalesced rows of M) x a range of blocks, 4 waves x 8 loads in flight.
__global__ __launch_bounds__(256) void knn_pool_collect_kernel(const float* __restrict__ M, const float* __restrict__ thrq, int B, int nblk,
                                                               int blk_per_wg, float* __restrict__ evalue, int* __restrict__ eblk,
                                                               int* __restrict__ cnt, int* __restrict__ flag) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q = blockIdx.y * 64 + lane;
  const bool qok = q < B;
  const int qc = qok ? q : B - 1;
  const float thr = thrq[qc];
  const int b0 = blockIdx.x * blk_per_wg;
  int b1 = b0 + blk_per_wg;
  b1 = b1 < nblk ? b1 : nblk;
  for (int b = b0 + w; b < b1; b += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (b + 4 * u < b1) ? M[(int64_t)(b + 4 * u) * B + qc] : -FLT_MAX;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (qok && b + 4 * u < b1 && v[u] >= thr) {
        const int pos = atomicAdd(cnt + q, 1);
        if (pos < POOL_CAP) { evalue[(int64_t)q * POOL_CAP + pos] = v[u]; eblk[(int64_t)q * POOL_CAP + pos] = b + 4 * u; }
        else atomicOr(flag, 1);
      }
  }
}

// Stage A re-scores the 16 rows of every surviving block APPROXIMATELY from the blocked bf16 copy (4 lanes per row, a quarter of the k
// range each; the block's rows sit in 64-byte runs of every cell: sector-efficient) — any bf16-product sum is within eps of the exact
// score, so a true top-k row still has s^' >= tau — and only the rows that pass go to stage B, the exact ascending-k fmaf chain over the
// fp32 row (1.5 - 3 KB scattered per row: re-ranking all 16 rows of every block exactly was 131 us of BASELINE configs[1]'s search and
// 0.7 ms at 1M x 768 x 1024 queries).
constexpr int POOL_EXACT = 512;        // rows re-ranked exactly per query
template <int D, bool FINE = false>
__global__ __launch_bounds__(256) void knn_pool_rerank_kernel(const float* __restrict__ q, const float* __restrict__ xb, const char* __restrict__ xblk,
                                                              int N, int k, const float* __restrict__ evalue, const int* __restrict__ eblk,
                                                              const int* __restrict__ cnt, const float* __restrict__ qnorm, float eps_scale,
                                                              float* __restrict__ dist, int64_t* __restrict__ idx, int* __restrict__ flag) {
  constexpr int KC = D / 8, CPL = KC / 4;                            // 16-byte k chunks per row / per lane of a row's quad
  __shared__ float sV[POOL_CAP];
  __shared__ int sB[POOL_CAP];
  constexpr int NSURV = FINE ? 256 : POOL_ROWS / 16;               // surviving blocks per query (4-row blocks: 1 024 rows; 16-row: 2 048)
  __shared__ int sSurv[NSURV];
  __shared__ float sS[POOL_EXACT];
  __shared__ int sI[POOL_EXACT];
  __shared__ int sN, sN2;
  __shared__ float sTau;
  __shared__ __attribute__((aligned(16))) float sQ[D];
  static_assert(D % 128 == 0, "knn_pool_rerank: D must be a multiple of 128");
  const int qg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = cnt[qg];
  if (n > POOL_CAP) { if (tid == 0) atomicOr(flag, 1); return; }     // (the collect kernel raised the flag already)
  if (tid == 0) { sN = 0; sN2 = 0; sTau = -FLT_MAX; }
  for (int d = tid; d < D; d += 256) sQ[d] = q[(int64_t)qg * D + d];
  for (int e = tid; e < n; e += 256) { sV[e] = evalue[(int64_t)qg * POOL_CAP + e]; sB[e] = eblk[(int64_t)qg * POOL_CAP + e]; }
  __syncthreads();
  // rank of an entry among the collected values (ties by block id): the entry of rank k - 1 is m_k
  for (int e = tid; e < n; e += 256) {
    const float v = sV[e]; const int b = sB[e];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (sV[j] > v || (sV[j] == v && sB[j] < b)) ? 1 : 0;
    const int kk = k < n ? k : n;                                      // fewer than k blocks in the whole index: everything survives
    if (rank == kk - 1) sTau = v - eps_scale * qnorm[qg];
  }
  __syncthreads();
  const float tau = sTau;
  for (int e = tid; e < n; e += 256)
    if (sV[e] >= tau) {
      const int pos = atomicAdd(&sN, 1);
      if (pos < NSURV) sSurv[pos] = sB[e];
    }
  __syncthreads();
  const int ns = sN;
  if (ns > NSURV) { if (tid == 0) atomicOr(flag, 1); return; }   // overflow: the gated exact pass recomputes everything
  const float* qr = q + (int64_t)qg * D;
  // ---- stage A: block bb = rows 64 (bb >> 2) + 32 ((bb >> 1) & 1) + 4 (bb & 1) + (r & 3) + 8 (r >> 2), r = 0..15; lane = (r, k quarter).
  // The lane's quarter of the query stays in registers; TWO blocks' chunks are requested before the first is summed.
  {
    const int r = lane >> 2, kq = lane & 3;
    // 16-row blocks: lane quad r covers row r of ONE block; 4-row blocks (FINE): the wave takes four blocks at a time, quad r = (block r >> 2, row r & 3)
    auto row_of = [&](int bb) __attribute__((always_inline)) {
      if constexpr (FINE) return 64 * (bb >> 4) + 32 * ((bb >> 3) & 1) + 4 * ((bb >> 2) & 1) + 8 * (bb & 3) + (r & 3);
      else return 64 * (bb >> 2) + 32 * ((bb >> 1) & 1) + 4 * (bb & 1) + (r & 3) + 8 * (r >> 2);
    };
    auto load_blk = [&](int row, u32x4 (&xv)[CPL]) __attribute__((always_inline)) {
      const char* cell = xblk + (((int64_t)(row >> 5) * KC + kq * CPL) * 32 + (row & 31)) * 16;
#pragma unroll
      for (int t = 0; t < CPL; ++t) xv[t] = *reinterpret_cast<const u32x4*>(cell + (size_t)t * 512);
    };
    auto finish = [&](int row, const u32x4 (&xv)[CPL]) __attribute__((always_inline)) {
      float acc = 0.f;
      int qo = 0;
      asm volatile("" : "+v"(qo));                         // opaque: else the query reads are hoisted out of the block loop (D / 4 registers)
      const float* qv = sQ + kq * CPL * 8 + qo;
#pragma unroll
      for (int t = 0; t < CPL; ++t) {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(qv + 8 * t), q1 = *reinterpret_cast<const f32x4*>(qv + 8 * t + 4);
        acc = fmaf(__uint_as_float(xv[t][0] << 16), q0[0], acc); acc = fmaf(__uint_as_float(xv[t][0] & 0xffff0000u), q0[1], acc);
        acc = fmaf(__uint_as_float(xv[t][1] << 16), q0[2], acc); acc = fmaf(__uint_as_float(xv[t][1] & 0xffff0000u), q0[3], acc);
        acc = fmaf(__uint_as_float(xv[t][2] << 16), q1[0], acc); acc = fmaf(__uint_as_float(xv[t][2] & 0xffff0000u), q1[1], acc);
        acc = fmaf(__uint_as_float(xv[t][3] << 16), q1[2], acc); acc = fmaf(__uint_as_float(xv[t][3] & 0xffff0000u), q1[3], acc);
      }
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      if (kq == 0 && row < N && acc >= tau) {
        const int pos = atomicAdd(&sN2, 1);
        if (pos < POOL_EXACT) sI[pos] = row;
      }
    };
    // (one block at a time: <= 128 registers keep four workgroups = sixteen queries per CU in flight, which hides the latency chain
    // cnt -> entries -> tau -> blocks -> rows better than a second block per wave did at two workgroups per CU)
    if constexpr (FINE) {
      for (int j = w * 4; j < ns; j += 16) {
        u32x4 xa[CPL];
        const int jj = j + (r >> 2);
        const int ra = jj < ns ? row_of(sSurv[jj]) : N;            // (past the survivors: row N is a zero row of the blocked copy or clamps below)
        load_blk(ra < N ? ra : N - 1, xa);
        finish(ra, xa);
      }
    } else
    for (int j = w; j < ns; j += 4) {
      u32x4 xa[CPL];
      const int ra = row_of(sSurv[j]);
      load_blk(ra, xa);
      finish(ra, xa);
    }
  }
  __syncthreads();
  const int n2 = sN2;
  if (n2 > POOL_EXACT) { if (tid == 0) atomicOr(flag, 1); return; }
  // ---- stage B: exact scores (the ascending-k fmaf chain) of the rows that passed.  One thread per row; the chain is serial but its
  // LOADS are not: 64 floats of the row are requested at once (an 8-load batch per 16 FMAs left the thread waiting out a memory round
  // trip per batch: 24 round trips per 384-wide row were most of this kernel), the query comes from LDS (broadcast reads).
  float scB[POOL_EXACT / 256]; int idB[POOL_EXACT / 256];
#pragma unroll
  for (int rr = 0; rr < POOL_EXACT / 256; ++rr) {
    float sc = -FLT_MAX; int id = ID_NONE;
    const int tix = tid + 256 * rr;
    if (tix < n2) {
      id = sI[tix];
      const float* xr = xb + (int64_t)id * D;
      sc = 0.f;
#pragma unroll 1
      for (int d = 0; d < D; d += 64) {
        f32x4 c[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u] = *reinterpret_cast<const f32x4*>(xr + d + 4 * u);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(sQ + d + 4 * u);
          sc = fmaf(a[0], c[u][0], sc); sc = fmaf(a[1], c[u][1], sc); sc = fmaf(a[2], c[u][2], sc); sc = fmaf(a[3], c[u][3], sc);
        }
      }
    }
    scB[rr] = sc; idB[rr] = id;
  }
  __syncthreads();                                                   // (sI is re-written below: every id has been read)
#pragma unroll
  for (int rr = 0; rr < POOL_EXACT / 256; ++rr) { sS[tid + 256 * rr] = scB[rr]; sI[tid + 256 * rr] = idB[rr]; }
  __syncthreads();
  if (tid >= 64) return;
  constexpr int PER = POOL_EXACT / 64;
  float ls[PER]; int li[PER];
#pragma unroll
  for (int t = 0; t < PER; ++t) { ls[t] = sS[lane + 64 * t]; li[t] = sI[lane + 64 * t]; }
  for (int o = 0; o < k; ++o) {
    float bs = -FLT_MAX; int bi = ID_NONE;
#pragma unroll
    for (int t = 0; t < PER; ++t)
      if (before(ls[t], li[t], bs, bi)) { bs = ls[t]; bi = li[t]; }
    float ws; int wi;
    wave_best(bs, bi, ws, wi);
    if (wi != ID_NONE) {
#pragma unroll
      for (int t = 0; t < PER; ++t)
        if (li[t] == wi) { ls[t] = -FLT_MAX; li[t] = ID_NONE; }
    }
    if (lane == 0) {
      dist[(int64_t)qg * k + o] = (wi == ID_NONE) ? -FLT_MAX : ws;
      idx[(int64_t)qg * k + o] = (wi == ID_NONE) ? (int64_t)-1 : (int64_t)wi;
    }
  }
}

int g_knn_qs_wgs = 0;                 // workgroups a knn_qs launch aims for (0 = one per CU); knn_set_option("qs_wgs")
bool g_knn_qs = true;                 // A/B switch (tests): 0 = the screened search ignores the blocked copy
int g_knn_qs_qt = 0;                  // A/B switch: query tiles per wave of the Q-stationary pass (0 = the rule in qs_plan, 1, 2)
struct QsPlan { int qt, nqg, ppc, nchunks; };
bool g_knn_qs_fine = true;            // A/B switch: 0 = 16-row blocks for every problem size
bool qs_fine(int64_t B, int64_t N) { return g_knn_qs_fine && N * B <= ((int64_t)64 << 20); }   // 4-row block maxima: N * B bytes
bool qs_applies(int64_t B, int64_t N, int D, int k) {
  return g_knn_qs && !g_knn_two_pass && !g_knn_force_tile && (D == 128 || D == 384 || D == 768) && k <= 16 && N >= 1024;   // (>= 16 pairs of 64 rows)
}
QsPlan qs_plan(int64_t B, int64_t N, int D) {
  QsPlan p;
  // two query tiles per wave (256 queries per workgroup: every A fragment read feeds two MFMAs — at one tile the four waves' reads run the
  // LDS at its full rate) where the index is long enough to keep every CU busy anyway; at D = 768 that is 384 fragment registers (500 of
  // 512 in all, no spill) and pays from 512 queries on at both widths (1M x 768 x 1024 q: 1.80 -> 1.54 ms, 256 q: 0.63 -> 0.87 ms; 1M x 384 x 256 q: 0.39 -> 0.66 ms); one tile for small
  // indexes, where more, shorter workgroups win
  p.qt = g_knn_qs_qt ? g_knn_qs_qt : ((N >= 65536 && B >= 512) ? 2 : 1);
  p.nqg = (int)((B + 128 * p.qt - 1) / (128 * p.qt));
  const int npairs = (int)((N + 63) / 64);
  int want = (g_knn_qs_wgs > 0 ? g_knn_qs_wgs : device_cus()) / p.nqg;
  if (want < 16) want = 16;                                // L(q) is the k-th largest CHUNK maximum (k <= 16): at least 16 chunks
  if (want > MAX_CHUNKS) want = MAX_CHUNKS;
  if (want > npairs) want = npairs;
  p.ppc = (npairs + want - 1) / want;
  p.nchunks = (npairs + p.ppc - 1) / p.ppc;
  if (p.nchunks < 16 && npairs <= MAX_CHUNKS) { p.ppc = 1; p.nchunks = npairs; }
  return p;
}
int launch_knn_qs_pool(const KnnArgs& a, int D, int qt, bool fine, hipStream_t s) {
  const dim3 grid((unsigned)(a.nqt * a.nchunks)), blk(256);
  if (fine) {                                              // (small problems only: one query tile per wave)
    if (D == 384) hipLaunchKernelGGL((knn_qs_kernel<384, 1, true>), grid, blk, 0, s, a);
    else if (D == 768) hipLaunchKernelGGL((knn_qs_kernel<768, 1, true>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((knn_qs_kernel<128, 1, true>), grid, blk, 0, s, a);
    return check_launch("knn_qs_pool");
  }
  if (D == 384 && qt == 2) hipLaunchKernelGGL((knn_qs_kernel<384, 2>), grid, blk, 0, s, a);
  else if (D == 384 && qt == 1) hipLaunchKernelGGL((knn_qs_kernel<384, 1>), grid, blk, 0, s, a);
  else if (D == 768 && qt == 1) hipLaunchKernelGGL((knn_qs_kernel<768, 1>), grid, blk, 0, s, a);
  else if (D == 768 && qt == 2) hipLaunchKernelGGL((knn_qs_kernel<768, 2>), grid, blk, 0, s, a);
  else if (D == 128 && qt == 2) hipLaunchKernelGGL((knn_qs_kernel<128, 2>), grid, blk, 0, s, a);
  else if (D == 128 && qt == 1) hipLaunchKernelGGL((knn_qs_kernel<128, 1>), grid, blk, 0, s, a);
  else return fail(EFFOCR_EINVAL, "knn(qs): internal");
  return check_launch("knn_qs_pool");
}

// how many queries one streaming launch takes at (D, kmax): 64 where both query images and the merge lists fit the LDS, else 32
int stream_queries(int D, int kmax) { return (D <= 384 && kmax <= 16) ? 64 : 32; }

}  // namespace

size_t knn_workspace_bytes(int64_t B, int64_t N, int D, int k) {
  (void)D;
  if (B <= 0 || k <= 0) return 0;
  const Plan p = make_plan(B, N, k);
  if (p.kmax == 0 || p.nchunks <= 1) return 256;
  return align_up((size_t)p.nchunks * (size_t)B * p.kmax * 8, 256) + 256;
}

int knn_ip_topk(const float* q, int64_t B, const float* xb, int64_t N, int D, int k, float* dist, int64_t* idx,
                void* ws, size_t ws_bytes, hipStream_t s) {
  if (B < 0 || N < 0 || D <= 0 || k <= 0) return fail(EFFOCR_EINVAL, "knn: bad sizes");
  if (B == 0) return EFFOCR_OK;
  if (D % 32 != 0) return fail(EFFOCR_EUNSUPPORTED, "knn: embedding dim must be a multiple of 32");
  if (N >= (int64_t)INT_MAX - 256 || B >= (int64_t)INT_MAX - 256) return fail(EFFOCR_EUNSUPPORTED, "knn: index or batch too large");
  const Plan p = make_plan(B, N, k);
  if (ws_bytes < knn_workspace_bytes(B, N, D, k)) return fail(EFFOCR_EWORKSPACE, "knn: workspace too small");
  KnnArgs a{};
  a.q = q; a.B = (int)B; a.xb = (N > 0) ? xb : q; a.N = (int)N; a.D = D; a.k = k; a.ldo = k; a.ocol = 0; a.after_col = -1;
  a.nqt = p.nqt; a.tiles_per_chunk = p.tpc; a.nchunks = p.nchunks;
  a.pdist = static_cast<float*>(ws);
  a.pidx = reinterpret_cast<int*>(static_cast<char*>(ws) + align_up((size_t)p.nchunks * (size_t)B * p.kmax * 4, 128));
  a.dist = dist; a.idx = idx;
  if (N == 0) { a.tiles_per_chunk = 0; a.nchunks = 1; }
  if (k > 32) {
    // faiss / PML accept any k (infer_effocr.py:317, viz_effocr_recognizer.py:78).  The register-resident lists hold 32 entries, so the
    // result is produced 32 columns per pass: pass p ranks only the rows that come strictly after pass p-1's last result in the
    // (score desc, id asc) order — every pass is the same exact scan, the concatenation is the exact sorted top-k.
    for (int done = 0; done < k; done += 32) {
      KnnArgs b = a;
      b.k = k - done < 32 ? k - done : 32;
      b.ocol = done; b.after_col = done > 0 ? done - 1 : -1;
      const int rc = launch_knn_k<float>(pick_kmax(b.k), b, s);
      if (rc) return rc;
    }
    return EFFOCR_OK;
  }
  // up to 128 queries against a large index: the streaming kernel (HBM / fp32-MFMA bound) in slices of 32 or 64 queries; same
  // chunking, same merge, same bits.  (Above that the 128-query tile kernel amortises the index traffic better.)
  const int sq = stream_queries(D, p.kmax);
  // (33..128 queries pay off only where the index does not fit the caches: at 10 k rows the tile kernel is faster, measured)
  if (B <= 2 * sq && N >= (B <= 32 ? 4096 : g_knn_stream_min_rows) && D % 128 == 0 && D <= 768 && !g_knn_force_tile) {
    for (int64_t q0 = 0; q0 < B; q0 += sq) {
      KnnArgs b = a;
      b.B = (int)(B - q0 < sq ? B - q0 : sq);
      b.q = q + q0 * D; b.dist = dist + q0 * k; b.idx = idx + q0 * k;
      const int rc = launch_knn_stream_k(p.kmax, b.B > 32 ? 2 : 1, b, s);   // (the partial lists are re-used: launches are stream-ordered)
      if (rc) return rc;
    }
    return EFFOCR_OK;
  }
  return launch_knn_k<float>(p.kmax, a, s);
}

// ---- screened search: bit-identical results to knn_ip_topk, for large indexes ------------------------------------------
// Workspace layout: [exact-pass partial lists | qb bf16 | qnorm | adist | aidx | cnt | flag | cand]
struct ScreenWs { size_t part, qb, qnorm, adist, aidx, cnt, flag, cand, pool_m, pool_c, pool_ev, pool_eb, total; };
ScreenWs screen_ws(int64_t B, int64_t N, int D, int k) {
  ScreenWs w; size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off = align_up(off + n, 256); return o; };
  size_t part = knn_workspace_bytes(B, N, D, k < 16 ? 16 : k);    // pass 1 keeps >= 16-entry chunk lists (see knn_ip_topk_screened)
  if (false) {
    const QsPlan qp = qs_plan(B, N, D);
    const size_t need = align_up((size_t)qp.nchunks * (size_t)B * 16 * 8, 256) + 256;
    part = part > need ? part : need;
  }
  w.part = take(part);
  w.qb = take((size_t)B * D * 2);
  w.qnorm = take((size_t)B * 4);
  w.adist = take((size_t)B * k * 4);
  w.aidx = take((size_t)B * k * 8);
  w.cnt = take((size_t)B * 4);
  w.flag = take(256);
  w.cand = take((size_t)B * RR_CAP * 4);
  w.pool_m = w.pool_c = w.pool_ev = w.pool_eb = 0;
  if (k <= 16 && N >= 1024 && (D == 128 || D == 384 || D == 768)) {   // block / chunk maxima + collected entries of the Q-stationary pass (only the widths it exists
                                                                      // for: a 1M x 512 index with 8 192 queries must not pay 2 GB for arrays no kernel writes; advisor, round 5)
    const QsPlan qp = qs_plan(B, N, D);
    w.pool_m = take((size_t)((N + 63) / 64) * (qs_fine(B, N) ? 16 : 4) * (size_t)B * 4);
    w.pool_c = take((size_t)qp.nchunks * QS_NSUB * (size_t)B * 4);
    w.pool_ev = take((size_t)B * POOL_CAP * 4);
    w.pool_eb = take((size_t)B * POOL_CAP * 4);
  }
  w.total = off;
  return w;
}
size_t knn_screen_workspace_bytes(int64_t B, int64_t N, int D, int k) {
  if (B <= 0 || k <= 0) return 0;
  return screen_ws(B, N, D, k).total;
}
void knn_force_tile_kernel(int on) { g_knn_force_tile = on != 0; }
void knn_two_pass_screen(int on) { g_knn_two_pass = on != 0; }
void knn_q16_tile(int on) { g_knn_q16 = on != 0; }
void knn_set_wg_target(int n) { g_knn_wg_target = n < 1 ? 1 : n; }
size_t knn_screen_flag_offset(int64_t B, int64_t N, int D, int k) {
  if (B <= 0 || k <= 0) return 0;
  return screen_ws(B, N, D, k).flag;
}

int convert_bf16(const float* src, int64_t n, void* dst, hipStream_t s) {
  if (n <= 0) return EFFOCR_OK;
  hipLaunchKernelGGL(convert_bf16_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, src, n, static_cast<__bf16*>(dst));
  return check_launch("convert_bf16");
}

int convert_bf16_blocked(const float* src, int64_t N, int D, void* dst, hipStream_t s) {
  if (N <= 0) return EFFOCR_OK;
  if (D % 8) return fail(EFFOCR_EUNSUPPORTED, "convert_bf16_blocked: the row length must be a multiple of 8");
  const int64_t slots = (N + 63) / 64 * 2 * (D / 8) * 32;
  hipLaunchKernelGGL(convert_bf16_blocked_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, s, src, N, D, static_cast<__bf16*>(dst));
  return check_launch("convert_bf16_blocked");
}
void knn_qs_option(int which, int value) {
  if (which == 0) g_knn_qs = value != 0;
  else if (which == 3) g_knn_stream_min_rows = value < 4096 ? 4096 : value;
  else if (which == 4) g_knn_qs_qt = (value == 1 || value == 2) ? value : 0;
  else if (which == 5) g_knn_qs_fine = value != 0;
  else g_knn_qs_wgs = value < 0 ? 0 : value;
}

int knn_ip_topk_screened(const float* q, int64_t B, const float* xb, const void* xb16, const void* xblk, int64_t N, int D, int k, float xnorm_max,
                         float* dist, int64_t* idx, void* ws, size_t ws_bytes, hipStream_t s) {
  if (B < 0 || N < 0 || D <= 0 || k <= 0 || !(xnorm_max >= 0.f)) return fail(EFFOCR_EINVAL, "knn(screened): bad sizes");
  if (B == 0) return EFFOCR_OK;
  if (D % 64 != 0) return fail(EFFOCR_EUNSUPPORTED, "knn(screened): embedding dim must be a multiple of 64");
  if (N < k) return fail(EFFOCR_EUNSUPPORTED, "knn(screened): needs at least k index rows (use the exact entry point)");
  if (N >= (int64_t)INT_MAX - 256 || B >= (int64_t)INT_MAX - 256) return fail(EFFOCR_EUNSUPPORTED, "knn: index or batch too large");
  const Plan p = make_plan(B, N, k);
  if (k > 32) return fail(EFFOCR_EUNSUPPORTED, "knn(screened): k > 32 runs on the exact multi-pass search (knn_ip_topk)");
  const ScreenWs w = screen_ws(B, N, D, k);
  if (ws_bytes < w.total) return fail(EFFOCR_EWORKSPACE, "knn(screened): workspace too small");
  char* W = static_cast<char*>(ws);
  __bf16* qb = reinterpret_cast<__bf16*>(W + w.qb);
  float* qnorm = reinterpret_cast<float*>(W + w.qnorm);
  float* adist = reinterpret_cast<float*>(W + w.adist);
  int64_t* aidx = reinterpret_cast<int64_t*>(W + w.aidx);
  int* cnt = reinterpret_cast<int*>(W + w.cnt);
  int* flag = reinterpret_cast<int*>(W + w.flag);
  int* cand = reinterpret_cast<int*>(W + w.cand);
  int rc = EFFOCR_OK;
  KnnArgs a{};
  a.B = (int)B; a.N = (int)N; a.D = D; a.k = k; a.ldo = k; a.ocol = 0; a.after_col = -1;
  a.nqt = p.nqt; a.tiles_per_chunk = p.tpc; a.nchunks = p.nchunks;
  a.pdist = reinterpret_cast<float*>(W + w.part);
  // pass 1: approximate top-k (only the k-th score is used)
  // Its per-chunk lists are the candidate source below, and "the list is FULL of qualifying rows" is the overflow signal — with
  // one-entry lists (k = 1: the driver's own call, infer_effocr_onnx_multi.py:372) the chunk that holds the approximate top-1 would
  // always look full and every search would also run the gated exact pass.  Lists of >= 16 entries make a full list mean what it says.
  const int kmax1 = (p.kmax < 16 && p.nchunks > 1 && !g_knn_two_pass) ? 16 : p.kmax;
  a.pidx = reinterpret_cast<int*>(W + w.part + align_up((size_t)p.nchunks * (size_t)B * kmax1 * 4, 128));
  a.q = qb; a.xb = xb16; a.dist = adist; a.idx = aidx;
  // 17..128 queries against a large index (at 256 the 128-query tile kernel is faster again: 1.0 vs 1.27 ms at 1M x 384; the ONNX driver's 64-crop calls against a jisx0213-scale index, infer_effocr_onnx_multi.py:372):
  // the screening pass STREAMS the bf16 index once per 64 queries (knn_stream_kernel<.., __bf16>) instead of running the 128-query tile
  // kernel; its chunk lists feed the same collect / re-rank / gated exact fallback.
  const float c = (0.00390625f + 0.0000152587890625f + 4.0f * (float)D * 5.9604645e-8f) * 1.0001f;   // |s^ - s| <= c |q| |x|, see pass 2
  const bool stream16 = B <= 128 && N >= 65536 && D % 192 == 0 && D <= 768 && p.nchunks > 1 && kmax1 >= 16 && !g_knn_two_pass && !g_knn_force_tile && xb16 != nullptr;
  // The Q-stationary pass over the fragment-blocked bf16 copy (knn_qs_kernel): every call size it covers except the 17..128-query calls
  // against a large index, which the streaming screen serves at the HBM rate.
  const bool qs = xblk != nullptr && qs_applies(B, N, D, k) && !stream16;
  // Neither screening pass can run (the caller handed over only the blocked copy and a process-global A/B option switched the Q-stationary
  // pass off): the exact search gives the same ids and score bits — never an error for a switch the caller cannot see (advisor, round 5).
  if (!qs && xb16 == nullptr) {
    if (hipMemsetAsync(flag, 0, 4, s) != hipSuccess) return fail(EFFOCR_EHIP, "knn(screened): clearing the overflow flag failed");
    return knn_ip_topk(q, B, xb, N, D, k, dist, idx, ws, ws_bytes, s);   // (its partial lists live in w.part, the front of the same workspace)
  }
  const QsPlan qsp = qs ? qs_plan(B, N, D) : QsPlan{};
  if (!qs) {                                               // (the Q-stationary chain rounds the queries itself and its bound kernel does the rest)
    hipLaunchKernelGGL(knn_prep_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, q, (int)B, D, qb, qnorm, cnt, flag);
    if ((rc = check_launch("knn_prep"))) return rc;
  }
  if (qs) {
    // pooled form (the default): pass 1 writes block / chunk maxima only, then threshold collect + block re-rank — four launches + the
    // gated exact pair, no lists, no merge
    KnnArgs b = a;
    b.q = q; b.xb = xblk; b.nqt = qsp.nqg; b.tiles_per_chunk = qsp.ppc; b.nchunks = qsp.nchunks;
    b.pool_m = reinterpret_cast<float*>(W + w.pool_m); b.pool_c = reinterpret_cast<float*>(W + w.pool_c);
    b.nsub = qsp.ppc < QS_NSUB ? qsp.ppc : QS_NSUB;
    const bool fine = qs_fine(B, N) && qsp.qt == 1;
    if ((rc = launch_knn_qs_pool(b, D, qsp.qt, fine, s))) return rc;
    const int nblk = (int)((N + 63) / 64) * (fine ? 16 : 4);
    // collect: ~2 workgroups per CU, 64 queries each
    const int nqb = (int)((B + 63) / 64);
    int per = (nblk * nqb + 2 * device_cus() - 1) / (2 * device_cus());
    per = (per + 31) / 32 * 32;
    if (per < 32) per = 32;
    const dim3 cg((unsigned)((nblk + per - 1) / per), (unsigned)nqb);
    float* ev = reinterpret_cast<float*>(W + w.pool_ev); int* eb = reinterpret_cast<int*>(W + w.pool_eb);
    const float es = 2.0f * c * xnorm_max;
    float* thr = reinterpret_cast<float*>(W + w.adist);     // (the approximate top-k area is free on this path)
    hipLaunchKernelGGL(knn_pool_bound_kernel, dim3((unsigned)B), dim3(256), 0, s, b.pool_c, (int)B, qsp.nchunks * b.nsub, k, q, D, qnorm, es, thr, cnt, flag);
    if ((rc = check_launch("knn_pool_bound"))) return rc;
    hipLaunchKernelGGL(knn_pool_collect_kernel, cg, dim3(256), 0, s, b.pool_m, thr, (int)B, nblk, per, ev, eb, cnt, flag);
    if ((rc = check_launch("knn_pool_collect"))) return rc;
    const char* xk = static_cast<const char*>(xblk);
    if (fine) {
      if (D == 384) hipLaunchKernelGGL((knn_pool_rerank_kernel<384, true>), dim3((unsigned)B), dim3(256), 0, s, q, xb, xk, (int)N, k, ev, eb, cnt, qnorm, es, dist, idx, flag);
      else if (D == 768) hipLaunchKernelGGL((knn_pool_rerank_kernel<768, true>), dim3((unsigned)B), dim3(256), 0, s, q, xb, xk, (int)N, k, ev, eb, cnt, qnorm, es, dist, idx, flag);
      else hipLaunchKernelGGL((knn_pool_rerank_kernel<128, true>), dim3((unsigned)B), dim3(256), 0, s, q, xb, xk, (int)N, k, ev, eb, cnt, qnorm, es, dist, idx, flag);
    } else
    if (D == 384) hipLaunchKernelGGL((knn_pool_rerank_kernel<384>), dim3((unsigned)B), dim3(256), 0, s, q, xb, xk, (int)N, k, ev, eb, cnt, qnorm, es, dist, idx, flag);
    else if (D == 768) hipLaunchKernelGGL((knn_pool_rerank_kernel<768>), dim3((unsigned)B), dim3(256), 0, s, q, xb, xk, (int)N, k, ev, eb, cnt, qnorm, es, dist, idx, flag);
    else hipLaunchKernelGGL((knn_pool_rerank_kernel<128>), dim3((unsigned)B), dim3(256), 0, s, q, xb, xk, (int)N, k, ev, eb, cnt, qnorm, es, dist, idx, flag);
    if ((rc = check_launch("knn_pool_rerank"))) return rc;
    KnnArgs e{};
    e.q = q; e.B = (int)B; e.xb = xb; e.N = (int)N; e.D = D; e.k = k; e.ldo = k; e.ocol = 0; e.after_col = -1;
    e.nqt = p.nqt; e.tiles_per_chunk = p.tpc; e.nchunks = p.nchunks;
    e.pdist = a.pdist; e.dist = dist; e.idx = idx; e.run_flag = flag;
    e.pidx = reinterpret_cast<int*>(W + w.part + align_up((size_t)p.nchunks * (size_t)B * p.kmax * 4, 128));
    // small indexes: the (gated, normally no-op) exact pass as ONE chunk per query tile — it writes the results itself, so the chain
    // loses the no-op merge launch (~5 us of a 50 us search at the reference's call sizes); a real overflow then costs a few ms once
    if (N <= 32768) { e.tiles_per_chunk = p.ntiles; e.nchunks = 1; }
    return launch_knn_k<float>(p.kmax, e, s);
  }
  if (stream16) {
    a.qnorm = qnorm;
    a.eps_scale = 2.0f * c * xnorm_max;                    // the band of the re-rank's candidate set (below)
    const int sq = kmax1 == 16 ? 64 : 32;                  // queries per launch: two query tiles where the lists hold 16 entries
    for (int64_t q0 = 0; q0 < B; q0 += sq) {
      KnnArgs b = a;
      b.B = (int)(B - q0 < sq ? B - q0 : sq);
      b.q = qb + q0 * D; b.qnorm = qnorm + q0; b.pB = (int)B; b.pq0 = (int)q0;
      if (b.B > 32) rc = launch_knn_stream<16, 2, false, __bf16>(b, s, false);
      else rc = kmax1 == 16 ? launch_knn_stream<16, 1, false, __bf16>(b, s, false) : launch_knn_stream<32, 1, false, __bf16>(b, s, false);
      if (rc) return rc;
    }
    if (kmax1 == 16) launch_knn_merge<16>(a.pdist, a.pidx, (int)B, p.nchunks, k, adist, aidx, nullptr, k, 0, s);
    else launch_knn_merge<32>(a.pdist, a.pidx, (int)B, p.nchunks, k, adist, aidx, nullptr, k, 0, s);
    if ((rc = check_launch("knn_merge"))) return rc;
  } else
  if ((rc = launch_knn_k<__bf16>(kmax1, a, s))) return rc;
  // pass 2: every row whose approximate score is within 2*eps of the k-th approximate score.
  // |s^ - s| <= eps = c * |q| * |x|: operand rounding (2^-8 + 2^-16) plus fp32 accumulation of both chains (4 d 2^-24),
  // 1e-4 relative slack for the fp32 norms.  A true top-k row has s >= s_(k), hence s^ >= s_(k) - eps >= s^_(k) - 2 eps.
  a.adist = adist; a.qnorm = qnorm; a.eps_scale = 2.0f * c * xnorm_max; a.cand = cand; a.cnt = cnt; a.cap = RR_CAP;
  if (p.nchunks > 1 && !g_knn_two_pass) {                  // the candidates are already in pass 1's per-chunk lists
    const dim3 cg((unsigned)((B + 3) / 4));
    switch (kmax1) {
      case 16: hipLaunchKernelGGL((knn_collect_lists_kernel<16>), cg, dim3(256), 0, s, a.pdist, a.pidx, (int)B, p.nchunks, k, adist, qnorm, a.eps_scale, cand, cnt, RR_CAP, flag); break;
      default: hipLaunchKernelGGL((knn_collect_lists_kernel<32>), cg, dim3(256), 0, s, a.pdist, a.pidx, (int)B, p.nchunks, k, adist, qnorm, a.eps_scale, cand, cnt, RR_CAP, flag); break;
    }
  } else
  switch (p.kmax) {
    case 1: hipLaunchKernelGGL((knn_partial_kernel<1, __bf16, true>), dim3((unsigned)(a.nqt * a.nchunks)), dim3(256), 0, s, a); break;
    case 16: hipLaunchKernelGGL((knn_partial_kernel<16, __bf16, true>), dim3((unsigned)(a.nqt * a.nchunks)), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((knn_partial_kernel<32, __bf16, true>), dim3((unsigned)(a.nqt * a.nchunks)), dim3(256), 0, s, a); break;
  }
  if ((rc = check_launch("knn_collect"))) return rc;
  // pass 3: exact re-rank
  hipLaunchKernelGGL(knn_rerank_kernel, dim3((unsigned)B), dim3(256), 0, s, q, xb, D, k, cand, cnt, RR_CAP, dist, idx, flag);
  if ((rc = check_launch("knn_rerank"))) return rc;
  // fallback, gated on the device: the exact search over everything if any query overflowed its candidate list
  KnnArgs e{};
  e.q = q; e.B = (int)B; e.xb = xb; e.N = (int)N; e.D = D; e.k = k; e.ldo = k; e.ocol = 0; e.after_col = -1;
  e.nqt = p.nqt; e.tiles_per_chunk = p.tpc; e.nchunks = p.nchunks;
  e.pdist = a.pdist; e.dist = dist; e.idx = idx; e.run_flag = flag;
  e.pidx = reinterpret_cast<int*>(W + w.part + align_up((size_t)p.nchunks * (size_t)B * p.kmax * 4, 128));   // its own split of the list area
  return launch_knn_k<float>(p.kmax, e, s);
}

int l2_normalize_rows(const float* x, int64_t B, int D, float* y, hipStream_t s) {
  if (B <= 0) return EFFOCR_OK;
  hipLaunchKernelGGL(l2norm_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, x, B, D, y);
  return check_launch("l2_normalize");
}

int gather_rows(const float* src, const int64_t* rows, int64_t n, int D, float* dst, hipStream_t s) {
  if (n <= 0) return EFFOCR_OK;
  const int64_t total = n * D;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, rows, n, D, dst);
  return check_launch("gather_rows");
}

}  // namespace effocr

#ifdef KNN_STAMP
extern "C" int effocr_debug_knn_stamps(unsigned long long* out, int n) {
  const int m = effocr::KNN_STAMP_WGS * effocr::KNN_STAMP_N;
  if (n < m) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(effocr::knn_stamps), (size_t)m * 8) == hipSuccess ? 0 : -2;
}
#endif
