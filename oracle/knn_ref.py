"""Oracle for the k-NN half of the path: IndexFlatIP.search / FaissKNN semantics.

TEST INFRASTRUCTURE (see oracle/__init__.py).  "parity unpinned" by the reference itself.
Two independent restatements:
  * ``flat_ip_search``       — ctypes wrapper of oracle/flat_ip.c (fp32 ascending-k fmaf chain,
                               ties -> lower id); bit-exact target for the HIP kernel.
  * ``flat_ip_search_f64``   — numpy float64 ``Q @ X.T`` + stable argsort; independent check of
                               the C code (ids agree whenever the top-k margins exceed fp32 noise).
Reference call sites: infer_effocr.py:184-187,317-319; infer_effocr_onnx_multi.py:372-375.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
NEG = -3.4028234663852886e+38      # faiss pads missing results with numeric_limits<float>::lowest()


def build(force=False):
    so = os.path.join(_HERE, "liboracle_flat_ip.so")
    src = os.path.join(_HERE, "flat_ip.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle_flat_ip.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build())
        f32p, i64p, i64 = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64), ctypes.c_int64
        lib.flat_ip_search_f32.argtypes = [f32p, i64, f32p, i64, i64, i64, f32p, i64p]
        lib.flat_ip_search_f32.restype = ctypes.c_int
        lib.flat_ip_scores_f32.argtypes = [f32p, i64, f32p, i64, i64, f32p]
        lib.flat_ip_scores_f32.restype = None
        lib.l2_normalize_f32.argtypes = [f32p, i64, i64, f32p]
        lib.l2_normalize_f32.restype = None
        lib.flat_ip_num_threads.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def num_threads():
    return int(_lib().flat_ip_num_threads())


def flat_ip_search(q, xb, k):
    """-> (dist [B,k] f32 descending, idx [B,k] int64); k > ntotal pads (-FLT_MAX, -1)."""
    q, xb = _f32(q), _f32(xb)
    B, D = q.shape
    N = xb.shape[0]
    assert xb.ndim == 2 and (N == 0 or xb.shape[1] == D)
    dist = np.empty((B, k), np.float32)
    idx = np.empty((B, k), np.int64)
    rc = _lib().flat_ip_search_f32(_p(q, ctypes.c_float), B, _p(xb, ctypes.c_float), N, D, k,
                                   _p(dist, ctypes.c_float), _p(idx, ctypes.c_int64))
    if rc != 0:
        raise RuntimeError(f"flat_ip_search_f32 failed rc={rc}")
    return dist, idx


def flat_ip_scores(q, xb):
    q, xb = _f32(q), _f32(xb)
    s = np.empty((q.shape[0], xb.shape[0]), np.float32)
    _lib().flat_ip_scores_f32(_p(q, ctypes.c_float), q.shape[0], _p(xb, ctypes.c_float), xb.shape[0],
                              q.shape[1], _p(s, ctypes.c_float))
    return s


def l2_normalize(x):
    x = _f32(x)
    y = np.empty_like(x)
    _lib().l2_normalize_f32(_p(x, ctypes.c_float), x.shape[0], x.shape[1], _p(y, ctypes.c_float))
    return y


def flat_ip_search_f64(q, xb, k):
    """Independent float64 restatement: stable argsort of -(Q @ X^T) (ties -> lower id)."""
    q = np.asarray(q, np.float64)
    xb = np.asarray(xb, np.float64)
    B, N = q.shape[0], xb.shape[0]
    s = q @ xb.T if N else np.zeros((B, 0))
    order = np.argsort(-s, axis=1, kind="stable")[:, :k]
    dist = np.take_along_axis(s, order, axis=1)
    if k > N:
        order = np.concatenate([order, np.full((B, k - N), -1, np.int64)], axis=1)
        dist = np.concatenate([dist, np.full((B, k - N), NEG)], axis=1)
    return dist.astype(np.float32), order.astype(np.int64)


def remove_ids(xb, ids):
    """faiss IndexFlat.remove_ids: delete rows and compact, later rows shift down
    (infer_effocr.py:209-212 filters candidate_chars in the same order)."""
    keep = np.ones(xb.shape[0], bool)
    keep[np.asarray(ids, np.int64)] = False
    return xb[keep]


# ------------------------------------------------------------------------------------------------
# Summation orders faiss is free to use.  IndexFlatIP.search computes Q X^T with an sgemm for >= 20 queries and a SIMD scan
# below; neither pins the order in which the D products of a score are added (BLAS blocks K by its micro-kernel's unroll and
# vector width; the scan keeps 4-16 lane partial sums), and whether a product is fused into the add (FMA) depends on the build.
# ``scores_in_order`` restates those families in true fp32 so that a test can ask: for which top-1 margin does the id depend
# on the order?  (tests/test_gpu_knn.py::test_top1_is_invariant_under_summation_order)
SUM_ORDERS = ("ascending_fma", "ascending", "reversed", "blocked8", "blocked16", "blocked32", "lanes8", "lanes16", "pairwise", "fp64")


def _pairwise_sum(parts):
    """Balanced binary tree over axis 0 in fp32."""
    parts = list(parts)
    while len(parts) > 1:
        nxt = [parts[i] + parts[i + 1] for i in range(0, len(parts) - 1, 2)]
        if len(parts) & 1:
            nxt.append(parts[-1])
        parts = nxt
    return parts[0]


def scores_in_order(q, xb, order):
    """[B,N] inner products accumulated in fp32 in the named order (``fp64``: float64, rounded once at the end)."""
    q, xb = _f32(q), _f32(xb)
    B, D = q.shape
    N = xb.shape[0]
    if order == "ascending_fma":
        return flat_ip_scores(q, xb)                       # the C oracle: fmaf chain, k ascending (what the HIP kernel is bit-exact with)
    if order == "fp64":
        return (q.astype(np.float64) @ xb.astype(np.float64).T).astype(np.float32)
    prod = lambda kk: q[:, kk, None] * xb[None, :, kk]     # fp32 product, rounded (no FMA)
    if order in ("ascending", "reversed"):
        acc = np.zeros((B, N), np.float32)
        for kk in (range(D) if order == "ascending" else range(D - 1, -1, -1)):
            acc = acc + prod(kk)
        return acc
    if order.startswith("blocked"):                        # sgemm-like: K cut into blocks, a block summed sequentially, block sums added in order
        w = int(order[7:])
        acc = np.zeros((B, N), np.float32)
        for k0 in range(0, D, w):
            part = np.zeros((B, N), np.float32)
            for kk in range(k0, min(D, k0 + w)):
                part = part + prod(kk)
            acc = acc + part
        return acc
    if order.startswith("lanes"):                          # SIMD scan: w lane accumulators (k mod w), reduced by a tree at the end
        w = int(order[5:])
        lanes = [np.zeros((B, N), np.float32) for _ in range(w)]
        for kk in range(D):
            lanes[kk % w] = lanes[kk % w] + prod(kk)
        return _pairwise_sum(lanes)
    if order == "pairwise":
        return _pairwise_sum([prod(kk) for kk in range(D)])
    raise ValueError(order)


def top1_in_order(q, xb, order):
    """argmax per query with the lowest-id tie rule, scores accumulated in ``order``."""
    s = scores_in_order(q, xb, order)
    return np.argmax(s, axis=1).astype(np.int64), s
