/* Oracle: C restatement of faiss.IndexFlatIP.search as used by the EffOCR recognizer.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) — never linked into the product library.
 * "parity unpinned": faiss is an un-vendored, un-pinned dependency of the reference (not even
 * listed in requirements.txt; README.md:23-27) and cannot be installed offline, so this file
 * restates its published algorithm and anchors on the reference's call sites:
 *   infer_effocr.py:184-187,317      FaissKNN(index_init_fn=faiss.IndexFlatIP) ... knn_func(emb, k=10)
 *   infer_effocr_onnx_multi.py:372    knn_func(embedding, k=1)
 *   train_effocr_recognizer.py:47-52  index rows = L2-normalised embeddings, row i <-> ref.txt line i
 *
 * IndexFlatIP.search(q, k): S = Q . X^T in fp32; per query the k largest scores sorted in
 * descending order, labels = row numbers; when k > ntotal the tail is label -1 with score
 * -FLT_MAX (faiss CMin<float>::neutral() == numeric_limits<float>::lowest()).
 *
 * faiss leaves two things unspecified, and this restatement DEFINES them (DESIGN.md "k-NN"):
 *   1. summation order of the dot product (BLAS sgemm for >= 20 queries, SIMD scan below):
 *      here score = fmaf(q[D-1], x[D-1], ... fmaf(q[1], x[1], fmaf(q[0], x[0], 0))) — one
 *      ascending-k chain of fused multiply-adds, one rounding per term.  This is exactly what
 *      gfx950's v_mfma_f32_32x32x2_f32 computes, so the HIP kernel is bit-identical.
 *   2. order among exactly equal scores: the lower row id ranks first.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ranks (s1,i1) strictly before (s2,i2)? */
static inline int before(float s1, int64_t i1, float s2, int64_t i2) {
    return (s1 > s2) || (s1 == s2 && i1 < i2);
}

/* scores[n] for one query, ascending-k fmaf chain; 4 rows interleaved to hide FMA latency
 * (the chains are independent, so interleaving does not change any result bit). */
static void score_rows(const float* q, const float* xb, int64_t n0, int64_t n1, int64_t D, float* out) {
    int64_t n = n0;
    for (; n + 4 <= n1; n += 4) {
        const float *x0 = xb + n * D, *x1 = x0 + D, *x2 = x1 + D, *x3 = x2 + D;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int64_t d = 0; d < D; ++d) {
            const float qd = q[d];
            a0 = fmaf(qd, x0[d], a0);
            a1 = fmaf(qd, x1[d], a1);
            a2 = fmaf(qd, x2[d], a2);
            a3 = fmaf(qd, x3[d], a3);
        }
        out[n - n0] = a0; out[n - n0 + 1] = a1; out[n - n0 + 2] = a2; out[n - n0 + 3] = a3;
    }
    for (; n < n1; ++n) {
        const float* x = xb + n * D;
        float a = 0.f;
        for (int64_t d = 0; d < D; ++d) a = fmaf(q[d], x[d], a);
        out[n - n0] = a;
    }
}

/* Exported: the full [B,N] score matrix (used by tests that check raw scores bit-for-bit). */
void flat_ip_scores_f32(const float* q, int64_t B, const float* xb, int64_t N, int64_t D, float* scores) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) score_rows(q + b * D, xb, 0, N, D, scores + b * N);
}

/* Exported: IndexFlatIP.search.  dist [B,k] fp32, idx [B,k] int64. */
int flat_ip_search_f32(const float* q, int64_t B, const float* xb, int64_t N, int64_t D, int64_t k,
                       float* dist, int64_t* idx) {
    if (B < 0 || N < 0 || D <= 0 || k <= 0) return -1;
    enum { CH = 1024 };
    int err = 0;
#pragma omp parallel
    {
        float* buf = (float*)malloc(sizeof(float) * CH);
        if (!buf) {
#pragma omp atomic write
            err = 1;
        }
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            if (!buf) continue;
            float* ds = dist + b * k;
            int64_t* is = idx + b * k;
            for (int64_t j = 0; j < k; ++j) { ds[j] = -FLT_MAX; is[j] = -1; }
            int64_t filled = 0;
            for (int64_t n0 = 0; n0 < N; n0 += CH) {
                const int64_t n1 = (n0 + CH < N) ? n0 + CH : N;
                score_rows(q + b * D, xb, n0, n1, D, buf);
                for (int64_t n = n0; n < n1; ++n) {
                    const float s = buf[n - n0];
                    if (filled == k && !before(s, n, ds[k - 1], is[k - 1])) continue;
                    int64_t pos = (filled < k) ? filled : k - 1;   /* slot to overwrite */
                    while (pos > 0 && before(s, n, ds[pos - 1], is[pos - 1])) {
                        ds[pos] = ds[pos - 1]; is[pos] = is[pos - 1]; --pos;
                    }
                    ds[pos] = s; is[pos] = n;
                    if (filled < k) ++filled;
                }
            }
        }
        free(buf);
    }
    return err ? -2 : 0;
}

/* torch.nn.functional.normalize(x, p=2, dim=1): y = x / max(||x||_2, 1e-12)  (infer_effocr.py:316).
 * Norm accumulated in double so that the oracle is independent of the kernel's reduction tree. */
void l2_normalize_f32(const float* x, int64_t B, int64_t D, float* y) {
    for (int64_t b = 0; b < B; ++b) {
        double ss = 0.0;
        for (int64_t d = 0; d < D; ++d) ss += (double)x[b * D + d] * (double)x[b * D + d];
        float nrm = (float)sqrt(ss);
        if (nrm < 1e-12f) nrm = 1e-12f;
        for (int64_t d = 0; d < D; ++d) y[b * D + d] = x[b * D + d] / nrm;
    }
}

int flat_ip_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
