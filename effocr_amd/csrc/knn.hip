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

namespace effocr {
namespace {

using namespace tile128;

int g_knn_wg_target = 1024;          // workgroups a launch aims for (2 resident per CU); knn_set_option("wg_target")
bool g_knn_force_tile = false;        // A/B switch (tests): 1 = always the 128-query tile kernel
bool g_knn_q16 = true;                // A/B switch (tests): 0 = calls of <= 16 queries on the 32-wide tile as well
bool g_knn_two_pass = false;          // A/B switch (tests): 1 = screened search collects its candidates with a second scan of the index
constexpr int ID_NONE = INT_MAX;          // internal sentinel id (ranks after every real id)
constexpr int MAX_CHUNKS = 256;

__device__ __forceinline__ bool before(float s1, int i1, float s2, int i2) {
  return (s1 > s2) || (s1 == s2 && i1 < i2);
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
  int ring;                         // knn_stream_kernel: LDS-DMA stages per wave
  // k > 32: the result is produced 32 columns at a time.  Pass p writes columns [ocol, ocol + k) of the [B][ldo] outputs and
  // (AFTER) only ranks rows that come strictly AFTER the previous pass's last result (after_col) in the (score desc, id asc) order.
  int ldo, ocol, after_col;
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

// merge the per-chunk sorted lists of one query: one wave per query, lane L owns chunks L, L+64, ...
template <int KMAX>
__global__ __launch_bounds__(256) void knn_merge_kernel(const float* __restrict__ pdist, const int* __restrict__ pidx,
                                                        int B, int nchunks, int k, float* __restrict__ dist,
                                                        int64_t* __restrict__ idx, const int* __restrict__ run_flag, int ldo, int ocol) {
  if (run_flag != nullptr && *run_flag == 0) return;
  constexpr int LPL = MAX_CHUNKS / 64;
  const int lane = threadIdx.x & 63;
  const int qg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qg >= B) return;                                   // whole wave exits together
  int ptr[LPL];
#pragma unroll
  for (int c = 0; c < LPL; ++c) ptr[c] = 0;
  for (int o = 0; o < k; ++o) {
    float bs = -FLT_MAX; int bi = ID_NONE; int bc = -1;
    if (o < KMAX) {
#pragma unroll
      for (int c = 0; c < LPL; ++c) {
        const int chunk = lane + 64 * c;
        if (chunk < nchunks && ptr[c] < KMAX) {
          const int64_t off = ((int64_t)chunk * B + qg) * KMAX + ptr[c];
          const float s = pdist[off]; const int i = pidx[off];
          if (bc < 0 || before(s, i, bs, bi)) { bs = s; bi = i; bc = c; }
        }
      }
    }
    // wave argmax with the compound order; real ids are unique, so the owner is identifiable
    float ws = bs; int wi = bi;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float os = __shfl_xor(ws, off, 64);
      const int oi = __shfl_xor(wi, off, 64);
      if (before(os, oi, ws, wi)) { ws = os; wi = oi; }
    }
    if (wi != ID_NONE && bc >= 0 && bi == wi) {
#pragma unroll
      for (int c = 0; c < LPL; ++c) ptr[c] += (c == bc);
    }
    if (lane == 0) {
      dist[(int64_t)qg * ldo + ocol + o] = ws;
      idx[(int64_t)qg * ldo + ocol + o] = (wi == ID_NONE) ? (int64_t)-1 : (int64_t)wi;
    }
  }
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
  hipLaunchKernelGGL((knn_merge_kernel<KMAX>), dim3((unsigned)((a.B + 3) / 4)), dim3(256), 0, s,
                     a.pdist, a.pidx, a.B, a.nchunks, a.k, a.dist, a.idx, a.run_flag, a.ldo, a.ocol);
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
    float ws = bs; int wi = bi;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float os = __shfl_xor(ws, off, 64);
      const int oi = __shfl_xor(wi, off, 64);
      if (before(os, oi, ws, wi)) { ws = os; wi = oi; }
    }
    if (wi != ID_NONE) {
#pragma unroll
      for (int t = 0; t < PER; ++t)
        if (li[t] == wi) { ls[t] = -FLT_MAX; li[t] = ID_NONE; }   // ids are unique: exactly one owner
    }
    if (lane == 0) {
      dist[(int64_t)qg * k + o] = ws;
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
// Q16 (<= 16 queries; NQT = 1): the query tile is 16 wide and the products run on v_mfma_f32_16x16x4_f32 — half the matrix time per index
// byte of the 32-wide tile, which at <= 32 queries costs as much as the HBM stream itself (16 B/clk/CU); the instruction adds its four k
// products in ascending k with one rounding each (tools/ubench/mfma16_order.hip: 256 of 256 results bit-identical to the fmaf chain), so
// the scores stay those of oracle/flat_ip.c.  A: lane (row l & 15, k l >> 4) reads ITS 4 bytes of the transposed stage; B: LDS image
// [m][k 0..3][16 queries]; C: lane holds query l & 15, rows 4 (l >> 4) + r of the 16-row group: four partial lists per query and wave.
template <int KMAX, int NQT, bool Q16 = false>
__global__ __launch_bounds__(KS_THREADS, 1) void knn_stream_kernel(KnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int w = wave_id();
  const int chunk = blockIdx.x;
  const int D = a.D, nm = D / 4;
  const float* Q = static_cast<const float*>(a.q);
  const float* X = static_cast<const float*>(a.xb);
  f32x2* sQ = reinterpret_cast<f32x2*>(smem);                       // [NQT][nm][2][32]
  if constexpr (Q16) {
    float* sQ1 = reinterpret_cast<float*>(smem);                    // [D / 4][4][16]
    for (int id = tid; id < D * 16; id += KS_THREADS) {
      const int q = id & 15, kk = id >> 4;                          // kk = 4 m + kq
      sQ1[id] = q < a.B ? Q[(int64_t)q * D + kk] : 0.f;
    }
  } else {
    for (int id = tid; id < NQT * nm * 64; id += KS_THREADS) {
      const int qt = id / (nm * 64), rem = id - qt * nm * 64;
      const int q = qt * 32 + (rem & 31), h = (rem >> 5) & 1, m = rem >> 6;
      f32x2 v = {0.f, 0.f};
      if (q < a.B) { v[0] = Q[(int64_t)q * D + 4 * m + h]; v[1] = Q[(int64_t)q * D + 4 * m + 2 + h]; }
      sQ[id] = v;
    }
  }
  __syncthreads();

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
  constexpr int P = 4;                                              // stages in flight per wave, in REGISTERS (16 KB per wave, 128 KB per CU)
  const int nsl = D / 32;                                           // stages (k slabs) per row block; D % 128 == 0 -> P divides it
  constexpr int NBUF = NQT == 1 ? 2 : 1;                            // transpose stages per wave
  char* stg = smem + (size_t)D * 128 * NQT + (size_t)w * NBUF * STG;   // the wave's private transpose buffer
  // stream of this wave: stage t = (block t / nsl, slab t % nsl).  A stage is fetched by 4 fully coalesced 16-byte loads per
  // lane (lane -> row 8i + lane / 8, chunk lane % 8: 8 rows x 128 contiguous bytes per instruction), parked in registers
  // while P - 1 older stages are consumed, then transposed through the wave's LDS buffer into the row-per-lane MFMA layout.
  // The chunk position is XOR-swizzled with the row so that both the writes and the fragment reads spread over the banks.
  // Everything is wave-private: no workgroup barrier in the loop, and ordinary loads let the compiler count vmcnt itself.
  const int nblk = row_lo + w * 32 < row_hi ? (row_hi - row_lo - w * 32 + WSTEP - 1) / WSTEP : 0;
  const int nst = nblk * nsl;
  f32x4 rg[P][4];
  auto fetch = [&](f32x4 (&r)[4], int blk, int sl) __attribute__((always_inline)) {
    const int r0i = row_lo + w * 32 + blk * WSTEP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = r0i + 8 * i + (lane >> 3);
      row = row < a.N ? row : a.N - 1;                              // clamp: rows past the end are masked below
#if KNN_STREAM_NT
      r[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(X + (int64_t)row * D) + sl * 128 + ((lane & 7) << 4)));
#else
      r[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(X + (int64_t)row * D) + sl * 128 + ((lane & 7) << 4));
#endif
    }
  };
  int fb = 0, fs = 0;                                               // (block, slab) of the next stage to fetch
#pragma unroll
  for (int u = 0; u < P; ++u) {
    if (fb < nblk) fetch(rg[u], fb, fs);
    if (++fs == nsl) { fs = 0; ++fb; }
  }
  f32x16 acc[NQT];
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[qt][r] = 0.f;
  f32x4 acc16[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};    // Q16: row groups 0-15 / 16-31 of the block
  const int q16 = lane & 15, kq = lane >> 4;
  const float* qp16 = reinterpret_cast<const float*>(smem) + kq * 16 + q16;
  int sl = 0, r0 = row_lo + w * 32;
  const int wr = lane >> 3, wc = lane & 7;
  for (int t0 = 0; t0 < nst; t0 += P) {
#pragma unroll
    for (int u = 0; u < P; ++u) {
      char* buf = stg + (u % NBUF) * STG;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = 8 * i + wr;
        *reinterpret_cast<f32x4*>(buf + rl * 128 + ((wc ^ (rl & 7)) << 4)) = rg[u][i];
      }
      if (fb < nblk) fetch(rg[u], fb, fs);                           // the slot's next stage (P stages ahead)
      if (++fs == nsl) { fs = 0; ++fb; }
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
        const float thr = ls[0][KMAX - 1];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = r0 + 16 * g + 4 * kq + r;
            hits |= (n < a.N && acc16[g][r] > thr) ? (1u << (4 * g + r)) : 0u;
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
      } else {
      // C layout: col = query (r31), rows = index rows (r & 3) + 8 (r >> 2) + 4 half, ascending with r
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) {
        uint32_t hits = 0;
        const float thr = ls[qt][KMAX - 1];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = r0 + (r & 3) + 8 * (r >> 2) + 4 * half;
          hits |= (n < a.N && acc[qt][r] > thr) ? (1u << r) : 0u;
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
      }
      }
      sl = 0; r0 += WSTEP;
    }
  }

  // ---- merge the 16 partial lists of every query through LDS (the query image is dead)
  __syncthreads();
  constexpr int NSRC = (KS_THREADS / 64) * (Q16 ? 4 : 2);
  constexpr int QW = Q16 ? 16 : 32;                                 // queries per tile
  float* mS = reinterpret_cast<float*>(smem);                       // [QW * NQT][NSRC][KMAX]
  int* mI = reinterpret_cast<int*>(smem + QW * NQT * NSRC * KMAX * 4);
  if constexpr (Q16) {
    const int src = w * 4 + kq;
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
      mS[(q16 * NSRC + src) * KMAX + t] = ls[0][t];
      mI[(q16 * NSRC + src) * KMAX + t] = li[0][t];
    }
  } else {
    const int src = w * 2 + half;
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
      for (int t = 0; t < KMAX; ++t) {
        mS[((qt * 32 + r31) * NSRC + src) * KMAX + t] = ls[qt][t];
        mI[((qt * 32 + r31) * NSRC + src) * KMAX + t] = li[qt][t];
      }
  }
  __syncthreads();
  if (tid < QW * NQT && tid < a.B) {
    const float* s0 = mS + tid * NSRC * KMAX;
    const int* i0 = mI + tid * NSRC * KMAX;
    int ptr[NSRC];
#pragma unroll
    for (int c = 0; c < NSRC; ++c) ptr[c] = 0;
    const int nout = (a.nchunks == 1) ? a.k : KMAX;
    for (int o = 0; o < nout; ++o) {
      float bs = -FLT_MAX; int bi = ID_NONE; int bsrc = -1;
      if (o < KMAX) {
#pragma unroll
        for (int c = 0; c < NSRC; ++c) {
          if (ptr[c] < KMAX) {
            const float sc = s0[c * KMAX + ptr[c]]; const int ic = i0[c * KMAX + ptr[c]];
            if (bsrc < 0 || before(sc, ic, bs, bi)) { bs = sc; bi = ic; bsrc = c; }
          }
        }
#pragma unroll
        for (int c = 0; c < NSRC; ++c) ptr[c] += (c == bsrc);
      }
      if (a.nchunks == 1) {
        a.dist[(int64_t)tid * a.ldo + a.ocol + o] = bs;
        a.idx[(int64_t)tid * a.ldo + a.ocol + o] = (bi == ID_NONE) ? (int64_t)-1 : (int64_t)bi;
      } else {
        const int64_t off = ((int64_t)chunk * a.B + tid) * KMAX + o;
        a.pdist[off] = bs;
        a.pidx[off] = bi;
      }
    }
  }
}

template <int KMAX, int NQT, bool Q16 = false>
int launch_knn_stream(const KnnArgs& a_in, hipStream_t s) {
  KnnArgs a = a_in;
  constexpr int NBUF = NQT == 1 ? 2 : 1;
  const size_t q_bytes = (size_t)a.D * 128 * NQT, m_bytes = (size_t)32 * NQT * (KS_THREADS / 64) * 2 * KMAX * 8;
  const size_t s_bytes = q_bytes + (size_t)(KS_THREADS / 64) * NBUF * 4096;
  if (s_bytes > 160 * 1024 || m_bytes > 160 * 1024) return fail(EFFOCR_EUNSUPPORTED, "knn(stream): embedding dim / k too large for the LDS image");
  const size_t lds = s_bytes > m_bytes ? s_bytes : m_bytes;
  // per launch: the attribute belongs to the (function, device) pair and a process may search on several GPUs; the call is cheap
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_stream_kernel<KMAX, NQT, Q16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return fail(EFFOCR_EHIP, "knn(stream): hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
  hipLaunchKernelGGL((knn_stream_kernel<KMAX, NQT, Q16>), dim3((unsigned)a.nchunks), dim3(KS_THREADS), lds, s, a);
  int rc = check_launch("knn_stream");
  if (rc != EFFOCR_OK || a.nchunks == 1) return rc;
  hipLaunchKernelGGL((knn_merge_kernel<KMAX>), dim3((unsigned)((a.B + 3) / 4)), dim3(256), 0, s,
                     a.pdist, a.pidx, a.B, a.nchunks, a.k, a.dist, a.idx, a.run_flag, a.ldo, a.ocol);
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
  if (B <= 2 * sq && N >= (B <= 32 ? 4096 : 65536) && D % 128 == 0 && D <= 768 && !g_knn_force_tile) {
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
struct ScreenWs { size_t part, qb, qnorm, adist, aidx, cnt, flag, cand, total; };
ScreenWs screen_ws(int64_t B, int64_t N, int D, int k) {
  ScreenWs w; size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off = align_up(off + n, 256); return o; };
  w.part = take(knn_workspace_bytes(B, N, D, k < 16 ? 16 : k));   // pass 1 keeps >= 16-entry chunk lists (see knn_ip_topk_screened)
  w.qb = take((size_t)B * D * 2);
  w.qnorm = take((size_t)B * 4);
  w.adist = take((size_t)B * k * 4);
  w.aidx = take((size_t)B * k * 8);
  w.cnt = take((size_t)B * 4);
  w.flag = take(256);
  w.cand = take((size_t)B * RR_CAP * 4);
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

int knn_ip_topk_screened(const float* q, int64_t B, const float* xb, const void* xb16, int64_t N, int D, int k, float xnorm_max,
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
  hipLaunchKernelGGL(knn_prep_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, q, (int)B, D, qb, qnorm, cnt, flag);
  int rc = check_launch("knn_prep");
  if (rc) return rc;
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
  if ((rc = launch_knn_k<__bf16>(kmax1, a, s))) return rc;
  // pass 2: every row whose approximate score is within 2*eps of the k-th approximate score.
  // |s^ - s| <= eps = c * |q| * |x|: operand rounding (2^-8 + 2^-16) plus fp32 accumulation of both chains (4 d 2^-24),
  // 1e-4 relative slack for the fp32 norms.  A true top-k row has s >= s_(k), hence s^ >= s_(k) - eps >= s^_(k) - 2 eps.
  const float c = (0.00390625f + 0.0000152587890625f + 4.0f * (float)D * 5.9604645e-8f) * 1.0001f;
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
  e.pdist = a.pdist; e.pidx = a.pidx; e.dist = dist; e.idx = idx; e.run_flag = flag;
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
