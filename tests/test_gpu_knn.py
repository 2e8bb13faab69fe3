"""-m gpu: the inner-product top-k kernel against the C oracle (oracle/flat_ip.c) — BIT-EXACT scores
and identical ids (ascending-k fmaf chain, ties -> lower id) — plus FaissKNN / IndexFlatIP semantics."""
import numpy as np
import pytest
import torch

from effocr_amd import _lib
from oracle import knn_ref

pytestmark = pytest.mark.gpu


def unit(a):
    return (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)


def make(B, N, D, seed):
    rng = np.random.default_rng(seed)
    X = unit(rng.standard_normal((N, D)))
    if N:
        Q = unit(X[rng.integers(0, N, B)] + 0.1 * rng.standard_normal((B, D)).astype(np.float32))
    else:
        Q = unit(rng.standard_normal((B, D)))
    return Q, X


@pytest.mark.parametrize("B,N,D,k", [
    (64, 96, 512, 10),          # BASELINE config 1
    (1024, 10000, 384, 10),     # BASELINE config 2
    (64, 10000, 384, 1),        # ONNX driver call (k=1, batches of 64)
    (7, 1000, 384, 10), (130, 257, 768, 16), (33, 129, 128, 32), (1, 1, 32, 1),
    (5, 5, 64, 10),             # k > ntotal -> (-FLT_MAX, -1) padding
    (300, 70000, 384, 10),      # multi-tile chunks + chunk merge
])
def test_knn_bit_exact(dev, B, N, D, k):
    from effocr_amd.knn import IndexFlatIP
    Q, X = make(B, N, D, seed=B + N + D + k)
    idx = IndexFlatIP(D, device=dev)
    idx.add(X)
    assert idx.ntotal == N
    Dg, Ig = idx.search(Q, k)
    Dr, Ir = knn_ref.flat_ip_search(Q, X, k)
    assert Ig.dtype == np.int64 and Dg.dtype == np.float32
    np.testing.assert_array_equal(Ig, Ir)
    np.testing.assert_array_equal(Dg.view(np.uint32), Dr.view(np.uint32))     # bit-exact scores


def test_knn_exact_ties_lowest_id(dev):
    from effocr_amd.knn import IndexFlatIP
    rng = np.random.default_rng(5)
    base = unit(rng.standard_normal((40, 384)))
    X = np.concatenate([base, base[:20], base[5:9], base])          # many exact duplicates
    Q = base[:33].copy()
    idx = IndexFlatIP(384, device=dev)
    idx.add(X)
    Dg, Ig = idx.search(Q, 10)
    Dr, Ir = knn_ref.flat_ip_search(Q, X, 10)
    np.testing.assert_array_equal(Ig, Ir)
    np.testing.assert_array_equal(Dg.view(np.uint32), Dr.view(np.uint32))
    for b in range(33):                                             # duplicates appear in id order
        same = Ig[b][Dg[b] == Dg[b][0]]
        assert list(same) == sorted(same)


def test_empty_index_and_empty_query(dev):
    from effocr_amd.knn import IndexFlatIP
    idx = IndexFlatIP(64, device=dev)
    D, I = idx.search(np.ones((3, 64), np.float32), 4)
    assert (I == -1).all() and (D == np.float32(knn_ref.NEG)).all()
    idx.add(np.eye(64, dtype=np.float32))
    D, I = idx.search(np.zeros((0, 64), np.float32), 4)
    assert D.shape == (0, 4) and I.shape == (0, 4)


def test_remove_ids_compacts_like_faiss(dev):
    from effocr_amd.knn import IndexFlatIP
    Q, X = make(50, 300, 384, seed=3)
    idx = IndexFlatIP(384, device=dev)
    idx.add(X)
    rm = np.array([0, 17, 299, 150, 17], dtype=np.int64)
    assert idx.remove_ids(rm) == 4 and idx.ntotal == 296
    Xc = knn_ref.remove_ids(X, rm)
    np.testing.assert_array_equal(idx.reconstruct_n(), Xc)
    Dg, Ig = idx.search(Q, 10)
    Dr, Ir = knn_ref.flat_ip_search(Q, Xc, 10)
    np.testing.assert_array_equal(Ig, Ir)


def test_faissknn_and_index_file_roundtrip(dev, tmp_path):
    from effocr_amd.knn import FaissKNN, IndexFlatIP, read_index
    Q, X = make(40, 500, 384, seed=11)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False)
    knn.train(torch.from_numpy(X))
    q = torch.from_numpy(Q).to(dev)
    d, i = knn(q, k=10)
    assert d.device == q.device and i.dtype == torch.int64 and tuple(i.shape) == (40, 10)
    Dr, Ir = knn_ref.flat_ip_search(Q, X, 10)
    np.testing.assert_array_equal(i.cpu().numpy(), Ir)
    # CPU query tensor -> results come back on the CPU (FaissKNN returns on query.device)
    d2, i2 = knn(torch.from_numpy(Q), k=10)
    assert d2.device.type == "cpu" and torch.equal(i2, i.cpu())
    # ref_includes_query drops the self match
    d3, i3 = knn(torch.from_numpy(X[:8]), k=3, ref_includes_query=True)
    _, Ir4 = knn_ref.flat_ip_search(X[:8], X, 4)
    np.testing.assert_array_equal(i3.numpy(), Ir4[:, 1:])
    p = tmp_path / "ref.index"
    knn.save(str(p))
    knn2 = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False)
    knn2.load(str(p))
    assert knn2.index.ntotal == 500 and knn2.index.d == 384
    np.testing.assert_array_equal(knn2.index.reconstruct_n(), X)
    _, i4 = knn2(q, k=10)
    assert torch.equal(i4, i)
    d33, i33 = knn2(q, k=33)                # any k (faiss semantics): above 32 the exact search runs in passes of 32 columns
    assert torch.equal(i33[:, :10], i) and i33.shape == (q.shape[0], 33)


def test_l2_normalize_matches_oracle(dev):
    from effocr_amd.knn import l2_normalize
    rng = np.random.default_rng(0)
    x = rng.standard_normal((77, 384)).astype(np.float32) * 5
    x[3] = 0                                 # zero row: x / max(0, 1e-12) = 0
    y = l2_normalize(torch.from_numpy(x).to(dev)).cpu().numpy()
    ref = knn_ref.l2_normalize(x)
    np.testing.assert_allclose(y, ref, rtol=2e-6, atol=1e-7)
    assert (y[3] == 0).all()


# ---------------------------------------------------------------- screened search (large indexes): bit-identical to the exact kernel
def _both(dev, X, Q, k, use_qs=True):
    from effocr_amd.knn import IndexFlatIP
    ex = IndexFlatIP(X.shape[1], device=dev, screen=False)
    sc = IndexFlatIP(X.shape[1], device=dev, screen=True)
    sc.use_qs = use_qs                                      # False: the screening kernels over the row-major bf16 copy (A/B switches of those)
    ex.add(X); sc.add(X)
    De, Ie = ex.search_device(Q, k)
    Ds, Is = sc.search_device(Q, k)
    torch.cuda.synchronize()
    return De, Ie, Ds, Is


@pytest.mark.parametrize("N,D,B,k", [(200_000, 128, 300, 10), (100_000, 768, 64, 1), (70_000, 384, 1024, 32), (3000, 64, 5, 16)])
def test_screened_search_is_bit_identical(dev, N, D, B, k):
    g = torch.Generator(device=dev).manual_seed(N + D)
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
    pick = torch.randint(0, N, (B,), generator=g, device=dev)
    Q = torch.nn.functional.normalize(X[pick] + 0.05 * torch.randn(B, D, generator=g, device=dev), dim=1)
    De, Ie, Ds, Is = _both(dev, X, Q, k)
    assert torch.equal(Ie, Is)
    assert torch.equal(De.view(torch.int32), Ds.view(torch.int32))
    assert (Ie[:, 0] == pick).all()
    # the default collects its candidates from pass 1's per-chunk lists; the second-scan variant (A/B switch) must agree bit for bit
    L = _lib.lib()
    _lib.check(L.effocr_knn_set_option(b"two_pass_screen", 1), "knn_set_option")
    try:
        _, _, D2, I2 = _both(dev, X, Q, k, use_qs=False)
    finally:
        _lib.check(L.effocr_knn_set_option(b"two_pass_screen", 0), "knn_set_option")
    assert torch.equal(I2, Is) and torch.equal(D2.view(torch.int32), Ds.view(torch.int32))


@pytest.mark.parametrize("N,D,B,k", [(150_000, 384, 64, 10), (150_000, 384, 17, 1), (100_000, 768, 100, 10), (80_000, 384, 128, 20),
                                     (80_000, 192, 33, 32)])
def test_streaming_screen_is_bit_identical(dev, N, D, B, k):
    """17..128 queries against a large index whose dim is a multiple of 192: the screening pass is the bf16 STREAMING kernel (64 queries per
    launch; 32 where the lists hold 32 entries) instead of the 128-query tile kernel.  Same guarantee: ids and scores bit for bit those
    of the exact search, no overflow flag on benign data; near-duplicate clusters and exact duplicates (tie rule) included."""
    g = torch.Generator(device=dev).manual_seed(N + D + B)
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
    X[7000:7003] = X[50]                                    # exact duplicates: ascending id
    X[30000:30008] = torch.nn.functional.normalize(X[30000:30001] + 2e-3 * torch.randn(8, D, generator=g, device=dev), dim=1)   # a tight cluster (fits a chunk list)
    pick = torch.randint(0, N, (B,), generator=g, device=dev)
    pick[0], pick[1] = 50, 30000
    Q = torch.nn.functional.normalize(X[pick] + 0.05 * torch.randn(B, D, generator=g, device=dev), dim=1)
    Q[0] = X[50]
    De, Ie, Ds, Is = _both(dev, X, Q, k)
    assert torch.equal(Ie, Is)
    assert torch.equal(De.view(torch.int32), Ds.view(torch.int32))
    assert Is[0, 0].item() == 50 and (k < 2 or Is[0, 1].item() == 7000)
    from effocr_amd.knn import IndexFlatIP
    sc = IndexFlatIP(D, device=dev, screen=True)
    sc.add(X)
    sc.search_device(Q, k)
    torch.cuda.synchronize()
    assert _screen_flag(sc, B, k) == 0                      # the candidate lists did not overflow: no exact fallback ran
    # the tile-kernel screen (A/B switch) agrees bit for bit
    L = _lib.lib()
    _lib.check(L.effocr_knn_set_option(b"force_tile", 1), "knn_set_option")
    try:
        _, _, D2, I2 = _both(dev, X, Q, k, use_qs=False)
    finally:
        _lib.check(L.effocr_knn_set_option(b"force_tile", 0), "knn_set_option")
    assert torch.equal(I2, Is) and torch.equal(D2.view(torch.int32), Ds.view(torch.int32))


@pytest.mark.parametrize("N,D,B,k", [(10_000, 384, 1024, 10), (10_000, 384, 64, 10), (9_999, 384, 128, 1), (10_000, 384, 1139, 1),
                                     (70_001, 768, 300, 10), (5_000, 128, 40, 16), (300_000, 384, 513, 10), (1024, 384, 5, 10),
                                     (200_037, 768, 1024, 10), (1100, 128, 257, 3)])
def test_q_stationary_screen_is_bit_identical(dev, N, D, B, k):
    """Round 5: with the fragment-blocked bf16 copy of the index the screening pass is knn_qs_kernel (queries stationary in registers,
    index rows through an LDS-DMA ring) for EVERY index size — BASELINE configs[1]'s own 10 000 x 384 search at 1024 / 64 / 128 queries,
    run_effocr's k = 1 call, configs[3]-like 768-wide rows, row counts that are not multiples of 64, query counts that are not
    multiples of 32.  Ids and scores bit for bit those of the exact search AND of the round-4 screening kernels (A/B switch), exact
    duplicates in ascending id order, a tight cluster, no overflow flag on benign data."""
    from effocr_amd.knn import IndexFlatIP
    g = torch.Generator(device=dev).manual_seed(N + D + B)
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
    if N > 5000:
        X[4000:4003] = X[50]                                 # exact duplicates: ascending id
        X[3000:3008] = torch.nn.functional.normalize(X[3000:3001] + 2e-3 * torch.randn(8, D, generator=g, device=dev), dim=1)   # a tight cluster
    X[N - 1] = X[min(50, N - 2)]                           # ... and one in the last (partial) 64-row pair
    pick = torch.randint(0, N, (B,), generator=g, device=dev)
    pick[0] = min(50, N - 2)
    if B > 1:
        pick[1] = 3000 if N > 5000 else 1
    Q = torch.nn.functional.normalize(X[pick] + 0.05 * torch.randn(B, D, generator=g, device=dev), dim=1)
    Q[0] = X[min(50, N - 2)]
    ex = IndexFlatIP(D, device=dev, screen=False)
    ex.add(X)
    De, Ie = ex.search_device(Q, k)
    sc = IndexFlatIP(D, device=dev, screen=True)
    sc.add(X)
    Ds, Is = sc.search_device(Q, k)
    torch.cuda.synchronize()
    assert sc._xblk is not None and sc._qs_ok(k)
    assert torch.equal(Ie, Is) and torch.equal(De.view(torch.int32), Ds.view(torch.int32))
    assert Is[0, 0].item() == min(50, N - 2)
    if N >= 9_000:
        assert _screen_flag(sc, B, k) == 0                  # the candidate lists did not overflow: no exact fallback ran
    old = IndexFlatIP(D, device=dev, screen=True)
    old.use_qs = False                                      # the round-4 screening kernels over the row-major copy
    old.add(X)
    Do, Io = old.search_device(Q, k)
    assert old._xblk is None and torch.equal(Io, Is) and torch.equal(Do.view(torch.int32), Ds.view(torch.int32))
    # another chunking of the same launch: identical
    L = _lib.lib()
    for opt, val, back in ((b"qs_wgs", 37, 0), (b"qs_fine", 0, 1), (b"qs_qt", 1, 0), (b"qs_qt", 2, 0)):   # chunking / block granularity / query tiles per wave
        _lib.check(L.effocr_knn_set_option(opt, val), "knn_set_option")
        try:
            D3, I3 = sc.search_device(Q, k)
        finally:
            _lib.check(L.effocr_knn_set_option(opt, back), "knn_set_option")
        assert torch.equal(I3, Is) and torch.equal(D3.view(torch.int32), Ds.view(torch.int32)), opt


def test_nan_query_through_the_pooled_screen(dev):
    """A non-finite query (an f16 overflow upstream) must come back as (-FLT_MAX, -1) padding from the screened search too — never as a
    plausible id, never as a stale threshold's leftovers — and must not disturb its neighbours' results."""
    from effocr_amd.knn import IndexFlatIP
    g = torch.Generator(device=dev).manual_seed(5)
    N, D, B, k = 10_000, 384, 600, 10
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
    Q = torch.nn.functional.normalize(torch.randn(B, D, generator=g, device=dev), dim=1)
    ex = IndexFlatIP(D, device=dev, screen=False); ex.add(X)
    De, Ie = ex.search_device(Q, k)
    sc = IndexFlatIP(D, device=dev); sc.add(X)
    assert sc._use_screen(k, B) and sc._qs_ok(k)
    sc.search_device(Q, k)                                   # a clean call first: leaves thresholds behind in the workspace
    Qn = Q.clone()
    Qn[7] = float("nan")
    Qn[300, 5] = float("inf")
    Ds, Is = sc.search_device(Qn, k)
    ok = torch.ones(B, dtype=torch.bool, device=dev); ok[7] = False; ok[300] = False
    assert (Is[7] == -1).all() and (Is[300] == -1).all()
    assert torch.equal(Is[ok], Ie[ok]) and torch.equal(Ds[ok].view(torch.int32), De[ok].view(torch.int32))


@pytest.mark.parametrize("N,B", [(12_000, 600), (90_000, 300)])
def test_pooled_screen_overflow_falls_back_exactly(dev, N, B):
    """More near-identical rows around some queries than the pooled chain's caps hold (512 collected blocks / 1 024-2 048 re-scored rows):
    the device flag un-gates the exact pass — one chunk per query tile on small indexes (no merge launch), the chunked form on large
    ones — and ids and scores are those of the exact search, duplicates in ascending id order."""
    from effocr_amd.knn import IndexFlatIP
    g = torch.Generator(device=dev).manual_seed(N)
    D, k = 384, 10
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
    centre = torch.nn.functional.normalize(torch.randn(1, D, generator=g, device=dev), dim=1)
    X[2000:5000] = torch.nn.functional.normalize(centre + 1e-4 * torch.randn(3000, D, generator=g, device=dev), dim=1)   # 3 000 near-identical rows
    X[7000:7003] = X[100]
    Q = torch.nn.functional.normalize(torch.randn(B, D, generator=g, device=dev), dim=1)
    Q[0], Q[1] = centre[0], X[100]
    ex = IndexFlatIP(D, device=dev, screen=False); ex.add(X)
    De, Ie = ex.search_device(Q, k)
    sc = IndexFlatIP(D, device=dev, screen=True); sc.add(X)
    Ds, Is = sc.search_device(Q, k)
    torch.cuda.synchronize()
    assert sc._xblk is not None and _screen_flag(sc, B, k) != 0
    assert torch.equal(Ie, Is) and torch.equal(De.view(torch.int32), Ds.view(torch.int32))
    assert Is[1, :4].tolist() == [100, 7000, 7001, 7002]


def test_large_query_batches_are_sliced(dev):
    """The pooled screen keeps ntotal / 16 * nq * 4 bytes of block maxima: query batches beyond IndexFlatIP.SCREEN_WS_BYTES of that are
    searched in slices — same results, bounded workspace."""
    from effocr_amd.knn import IndexFlatIP
    g = torch.Generator(device=dev).manual_seed(3)
    N, D, B, k = 70_000, 128, 1500, 10
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
    Q = torch.nn.functional.normalize(torch.randn(B, D, generator=g, device=dev), dim=1)
    ref = IndexFlatIP(D, device=dev, screen=True)
    ref.add(X)
    Dr, Ir = ref.search_device(Q, k)
    sl = IndexFlatIP(D, device=dev, screen=True)
    sl.SCREEN_WS_BYTES = N * 600 // 4                       # forces slices of 512 queries
    sl.add(X)
    Ds, Is = sl.search_device(Q, k)
    assert torch.equal(Ir, Is) and torch.equal(Dr.view(torch.int32), Ds.view(torch.int32))
    ws = sl._ws[torch.cuda.current_stream(dev).cuda_stream]
    assert ws.numel() < ref._ws[torch.cuda.current_stream(dev).cuda_stream].numel()


def test_screened_search_non_unit_rows_ties_and_overflow(dev):
    """Rows of very different norms (the bound uses the max norm), exact duplicates (tie rule: lower id first) and
    a cluster of > 512 near-identical rows around some queries (candidate overflow -> gated exact fallback)."""
    g = torch.Generator(device=dev).manual_seed(7)
    N, D = 90_000, 256
    X = torch.randn(N, D, generator=g, device=dev) * torch.rand(N, 1, generator=g, device=dev) * 3
    X[5000:5004] = X[100]                                   # exact duplicates of row 100
    centre = torch.nn.functional.normalize(torch.randn(1, D, generator=g, device=dev), dim=1) * 2.5
    X[20000:21500] = centre + 1e-4 * torch.randn(1500, D, generator=g, device=dev)     # 1500 near-identical rows
    Q = torch.cat([X[[100, 7, 5001]], centre, torch.randn(60, D, generator=g, device=dev)])
    for k in (1, 10):
        De, Ie, Ds, Is = _both(dev, X, Q, k)
        assert torch.equal(Ie, Is) and torch.equal(De.view(torch.int32), Ds.view(torch.int32))
    assert Is[0, 0].item() == 100 and Is[2, 0].item() == 100 and Is[0, 1].item() == 5000      # duplicates: ascending id


def _screen_flag(idx, n, k):
    """Device overflow flag of the index's last screened search (the gated exact pass ran iff it is non-zero)."""
    L = _lib.lib()
    need = int(L.effocr_knn_screen_workspace_bytes(n, idx.ntotal, idx.d, k))
    off = int(L.effocr_knn_screen_flag_offset(n, idx.ntotal, idx.d, k))
    ws = idx._workspace(need)
    torch.cuda.synchronize()
    return int(ws[off:off + 4].view(torch.int32).item())


@pytest.mark.parametrize("k", [1, 10, 16])
def test_screened_search_does_not_fall_back_on_benign_queries(dev, k):
    """k = 1 is the ONNX driver's own call (infer_effocr_onnx_multi.py:372) and run_effocr's: a benign screened search (every query a
    noisy copy of one row, no near-duplicate clusters) must finish WITHOUT raising the overflow flag — i.e. without also running the
    gated exact pass — and 'auto' screening must stay on over many calls."""
    from effocr_amd.knn import IndexFlatIP
    g = torch.Generator(device=dev).manual_seed(11 + k)
    N, D, B = 150_000, 384, 256
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
    pick = torch.randint(0, N, (B,), generator=g, device=dev)
    Q = torch.nn.functional.normalize(X[pick] + 0.05 * torch.randn(B, D, generator=g, device=dev), dim=1)
    idx = IndexFlatIP(D, device=dev, screen="auto")
    idx.add(X)
    assert idx._use_screen(k, B)
    ex = IndexFlatIP(D, device=dev, screen=False)
    ex.add(X)
    De, Ie = ex.search_device(Q, k)
    for _ in range(6):
        Ds, Is = idx.search_device(Q, k)
        assert _screen_flag(idx, B, k) == 0
        assert torch.equal(Ie, Is) and torch.equal(De.view(torch.int32), Ds.view(torch.int32))
    idx._poll_overflow()
    assert idx.screen == "auto" and idx.screen_overflows == 0


def test_screened_auto_threshold_and_invalidation(dev):
    from effocr_amd.knn import IndexFlatIP
    idx = IndexFlatIP(128, device=dev)
    assert not idx._use_screen(10)
    idx.add(torch.nn.functional.normalize(torch.randn(70_000, 128, device=dev), dim=1))
    assert idx._use_screen(10) and not idx._use_screen(33)
    # with the blocked copy's Q-stationary pass (d = 128 qualifies) the screened search wins from 33 queries on for k > 1; k = 1 keeps
    # the exact streaming kernel up to 64 queries (d <= 384); without it (use_qs = False) the round-4 rule: > 64 queries
    assert idx._use_screen(10, 33) and not idx._use_screen(10, 32)
    assert idx._use_screen(1, 65) and not idx._use_screen(1, 64)
    idx.use_qs = False
    assert idx._use_screen(10, 65) and not idx._use_screen(10, 64)
    idx.use_qs = True
    q = idx._xb[:140].clone()
    D1, I1 = idx.search_device(q, 5)
    assert (idx._xb16 is not None or idx._xblk is not None) and (I1[:, 0].cpu() == torch.arange(140)).all()
    q = q[:4]
    idx.remove_ids(np.array([0]))
    assert idx._xb16 is None and idx._xblk is None
    D2, I2 = idx.search_device(q[1:], 5)
    assert (I2[:, 0].cpu() == torch.arange(3)).all()        # rows shifted down by one


@pytest.mark.parametrize("B,N,D,k", [(1, 5000, 384, 10), (16, 70001, 384, 10), (9, 40000, 256, 1), (16, 30011, 128, 32), (2, 1000000, 384, 10), (32, 4096, 512, 1), (7, 200000, 768, 32), (32, 33333, 128, 16), (5, 4100, 1024, 10),
                                     (64, 70001, 384, 1), (33, 66000, 384, 10), (64, 70000, 384, 16), (100, 66001, 384, 10), (128, 65536, 256, 10),
                                     (64, 66000, 768, 10), (40, 70000, 384, 32), (100, 20000, 384, 10)])
                                     # 33..128 queries against >= 65 536 rows: two query tiles per launch / slices of 32 (below: the tile kernel)
def test_streaming_kernel_small_batches_bit_exact(hip_lib, dev, B, N, D, k):
    """<= 128 queries against >= 4096 rows run the streaming kernel (index rows straight into MFMA operands at the HBM rate):
    scores and ids bit-identical to the C oracle AND to the 128-query tile kernel (A/B switch), incl. planted exact ties."""
    from effocr_amd.knn import IndexFlatIP
    g = torch.Generator().manual_seed(B + N + D)
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1)
    X[N // 2] = X[3]
    X[N - 1] = X[3]                                         # three identical rows: ids 3 < N/2 < N-1 must come out in this order
    Q = torch.nn.functional.normalize(X[:B] + 0.05 * torch.randn(B, D, generator=g), dim=1)
    if B > 3:
        Q[3] = X[3]
    idx = IndexFlatIP(D, device=dev, screen=False)
    idx.add(X)
    Dv, Iv = idx.search_device(Q.to(dev), k)
    D_ref, I_ref = knn_ref.flat_ip_search(Q.numpy(), X.numpy(), k)
    assert np.array_equal(Iv.cpu().numpy(), I_ref)
    assert np.array_equal(Dv.cpu().numpy().view(np.uint32), D_ref.view(np.uint32))
    _lib.check(hip_lib.effocr_knn_set_option(b"force_tile", 1), "knn_set_option")
    try:
        Dt, It = idx.search_device(Q.to(dev), k)
    finally:
        _lib.check(hip_lib.effocr_knn_set_option(b"force_tile", 0), "knn_set_option")
    assert torch.equal(It, Iv) and torch.equal(Dt.view(torch.int32), Dv.view(torch.int32))
    if B <= 16:                                             # default: the 16-wide query tile (v_mfma_f32_16x16x4_f32); switch: the 32-wide one
        _lib.check(hip_lib.effocr_knn_set_option(b"q16_tile", 0), "knn_set_option")
        try:
            D32, I32 = idx.search_device(Q.to(dev), k)
        finally:
            _lib.check(hip_lib.effocr_knn_set_option(b"q16_tile", 1), "knn_set_option")
        assert torch.equal(I32, Iv) and torch.equal(D32.view(torch.int32), Dv.view(torch.int32))
    assert hip_lib.effocr_knn_set_option(b"nope", 1) == -1


def test_chunk_plan_does_not_change_the_result(hip_lib, dev):
    """The tile kernel cuts the index into chunks so that ~wg_target workgroups exist; every score is still one ascending-k
    fp32 chain and the merge of the partial lists is exact: ids and score bits are the same for any chunking and equal the oracle."""
    from effocr_amd.knn import IndexFlatIP
    g = torch.Generator().manual_seed(77)
    X = torch.nn.functional.normalize(torch.randn(10000, 384, generator=g), dim=1)
    Q = torch.nn.functional.normalize(torch.randn(200, 384, generator=g), dim=1)
    D_ref, I_ref = knn_ref.flat_ip_search(Q.numpy(), X.numpy(), 10)
    try:
        for target in (64, 512, 1024, 4096):
            _lib.check(hip_lib.effocr_knn_set_option(b"wg_target", target), "knn_set_option")
            idx = IndexFlatIP(384, device=dev, screen=False)
            idx.add(X)
            Dv, Iv = idx.search_device(Q.to(dev), 10)
            assert np.array_equal(Iv.cpu().numpy(), I_ref), target
            assert np.array_equal(Dv.cpu().numpy().view(np.uint32), D_ref.view(np.uint32)), target
    finally:
        _lib.check(hip_lib.effocr_knn_set_option(b"wg_target", 1024), "knn_set_option")


def planted_near_ties(D, margins, per_margin, seed):
    """Unit queries with TWO competing unit rows each: cosine 0.9 and 0.9 - margin (fp64 margin of the fp32-rounded rows is
    returned, it differs from the nominal one by ~1e-8), plus 400 unrelated rows."""
    rng = np.random.default_rng(seed)
    Q, rows, nominal = [], [], []
    for m in margins:
        for _ in range(per_margin):
            q = rng.standard_normal(D); q /= np.linalg.norm(q)
            uv = []
            for _ in range(2):
                u = rng.standard_normal(D); u -= (u @ q) * q; u /= np.linalg.norm(u)
                uv.append(u)
            c1, c2 = 0.9, 0.9 - m
            rows.append(c1 * q + np.sqrt(1 - c1 * c1) * uv[0])
            rows.append(c2 * q + np.sqrt(1 - c2 * c2) * uv[1])
            Q.append(q); nominal.append(m)
    Q = np.asarray(Q, np.float32)
    X = np.concatenate([np.asarray(rows, np.float32), unit(rng.standard_normal((400, D)))])
    perm = rng.permutation(X.shape[0])                     # the better row is not always the lower id
    X = X[perm]
    s64 = Q.astype(np.float64) @ X.astype(np.float64).T
    top2 = np.sort(s64, axis=1)[:, -2:]
    return Q, X, top2[:, 1] - top2[:, 0], np.asarray(nominal)


def test_top1_is_invariant_under_summation_order(dev, capsys):
    """north_star: "identical top-1 glyph IDs" vs faiss, whose sgemm / SIMD scan may add the D products of a score in any order.
    The HIP kernel is bit-exact with ONE order (ascending-k fmaf chain, oracle/flat_ip.c).  Here every other order family faiss
    could use (oracle/knn_ref.SUM_ORDERS: sequential with / without FMA, reversed, K-blocked by 8/16/32, 8/16 SIMD lane
    accumulators, pairwise tree, float64) must give the SAME top-1 id as the HIP kernel whenever the top-1 margin exceeds
    SAFE = 1e-6 in cosine (rigorous bound for unit rows: 2 * D * 2^-24 = 4.6e-5 at D = 384; measured divergence below ~2e-7),
    on the committed golden fixture and on planted near-ties; the margin at which orders start to disagree is printed."""
    from effocr_amd.knn import IndexFlatIP
    SAFE = 1e-6
    import os
    G = os.path.join(os.path.dirname(__file__), "golden")
    gold = np.load(os.path.join(G, "knn_c2small.npz"))
    cases = [("golden", gold["Q"].astype(np.float32), gold["X"].astype(np.float32))]
    margins = [1e-3, 1e-4, 1e-5, 3e-6, 1e-6, 3e-7, 1e-7, 3e-8, 1e-8, 0.0]
    Qp, Xp, true_margin, _ = planted_near_ties(384, margins, 24, seed=77)
    cases.append(("planted", Qp, Xp))
    Qr, Xr = make(256, 5000, 384, seed=99)                 # realistic: perturbed copies of index rows (margins ~0.1..0.5)
    cases.append(("random", Qr, Xr))
    worst_disagree = 0.0
    for name, Q, X in cases:
        idx = IndexFlatIP(Q.shape[1], device=dev)
        idx.add(X)
        _, Ig = idx.search(Q, 1)
        hip = Ig[:, 0]
        s64 = Q.astype(np.float64) @ X.astype(np.float64).T
        t2 = np.sort(s64, axis=1)[:, -2:]
        margin = t2[:, 1] - t2[:, 0]
        for order in knn_ref.SUM_ORDERS:
            ids, _ = knn_ref.top1_in_order(Q, X, order)
            differ = ids != hip
            assert not differ[margin > SAFE].any(), (name, order, margin[differ & (margin > SAFE)])
            if differ.any():
                worst_disagree = max(worst_disagree, float(margin[differ].max()))
        if name == "golden":                                # the fixture's margins are > 1e-4 by construction: every order agrees
            assert (margin > 1e-4).all()
    assert worst_disagree < SAFE
    with capsys.disabled():
        print(f"\n[knn] top-1 ids identical under {len(knn_ref.SUM_ORDERS)} summation orders for every margin > {SAFE:g}; "
              f"largest margin at which any order picked another row: {worst_disagree:.2e}")


@pytest.mark.parametrize("B,N,D,k", [(5, 300, 64, 50), (70, 5000, 384, 100), (130, 1000, 128, 33), (3, 40, 32, 64), (16, 70000, 384, 40)])
def test_k_above_32_multi_pass(dev, B, N, D, k):
    """faiss.IndexFlatIP.search / PML accept any k (infer_effocr.py:317, viz_effocr_recognizer.py:78).  Above 32 the result is
    produced 32 columns per pass (each pass the same exact scan, ranking only what comes after the previous pass's last result):
    ids and score bits equal the C oracle, including exact duplicates straddling a pass boundary and k > ntotal padding."""
    from effocr_amd.knn import FaissKNN, IndexFlatIP
    Q, X = make(B, N, D, seed=B + N + k)
    if N > 40:
        X[7] = X[5]; X[N - 1] = X[5]; X[N // 2] = X[5]          # identical rows
        Q[0] = X[5]
    idx = IndexFlatIP(D, device=dev)
    idx.add(X)
    Dg, Ig = idx.search(Q, k)
    Dr, Ir = knn_ref.flat_ip_search(Q, X, k)
    np.testing.assert_array_equal(Ig, Ir)
    np.testing.assert_array_equal(Dg.view(np.uint32), Dr.view(np.uint32))
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False, device=dev)
    knn.index = idx
    d, i = knn(torch.from_numpy(Q).to(dev), k=k)
    assert np.array_equal(i.cpu().numpy(), Ir)
