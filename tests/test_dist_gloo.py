"""-m "not gpu": the N>1 path (crop sharding + all_gather of the per-rank ids) on 2 CPU processes
with the gloo backend; the per-rank neighbour function is the CPU oracle here."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from effocr_amd.dist import ShardedRecognizer, shard_bounds
        from oracle import knn_ref
        rng = np.random.default_rng(0)
        X = rng.standard_normal((300, 64)).astype(np.float32)
        Q = rng.standard_normal((n, 64)).astype(np.float32)

        def neighbors(local):                      # stands in for Recognizer.neighbors on this rank's GPU
            d, i = knn_ref.flat_ip_search(local.numpy(), X, 5)
            return torch.from_numpy(d), torch.from_numpy(i)

        sharded = ShardedRecognizer(neighbors)
        d, i = sharded(torch.from_numpy(Q))
        lo, hi = shard_bounds(n, rank, world)
        d2, i2 = sharded(torch.from_numpy(Q[lo:hi]), n_total=n, presharded=True)
        assert torch.equal(i, i2) and torch.equal(d, d2)
        np.save(os.path.join(out_dir, f"i{rank}.npy"), i.numpy())
        np.save(os.path.join(out_dir, f"d{rank}.npy"), d.numpy())
    finally:
        dist.destroy_process_group()


def _worker8(rank, world, port, n, out_dir):
    """BASELINE configs[2]'s shape of the N > 1 path: 8 ranks, 1024 crops -> 128 per rank (and a ragged 1021), k = 10, and
    exactly ONE collective per call (DESIGN 5 / north_star: "an RCCL all-gather of the per-rank transcriptions")."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import effocr_amd.dist as ed
        from oracle import knn_ref
        rng = np.random.default_rng(1)
        X = rng.standard_normal((500, 32)).astype(np.float32)
        X[7] = X[3]                                           # a duplicate row: the ascending-id tie rule crosses the gather intact
        Q = rng.standard_normal((n, 32)).astype(np.float32)
        seen = []

        def neighbors(local):
            seen.append(len(local))
            d, i = knn_ref.flat_ip_search(local.numpy(), X, 10)
            return torch.from_numpy(d), torch.from_numpy(i)

        calls = {"n": 0}
        real_list, real_into = dist.all_gather, dist.all_gather_into_tensor

        def counted_list(*a, **k):
            calls["n"] += 1
            return real_list(*a, **k)

        def counted_into(*a, **k):
            calls["n"] += 1
            return real_into(*a, **k)

        dist.all_gather, dist.all_gather_into_tensor = counted_list, counted_into
        try:
            d, i = ed.ShardedRecognizer(neighbors)(torch.from_numpy(Q))
        finally:
            dist.all_gather, dist.all_gather_into_tensor = real_list, real_into
        lo, hi = ed.shard_bounds(n, rank, world)
        assert seen == [hi - lo] and calls["n"] == 1, (seen, calls)
        assert d.dtype == torch.float32 and i.dtype == torch.int64 and tuple(i.shape) == (n, 10)
        if rank in (0, world - 1):
            np.save(os.path.join(out_dir, f"i{rank}.npy"), i.numpy())
            np.save(os.path.join(out_dir, f"d{rank}.npy"), d.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1024, 1021, 5])
def test_sharded_recognizer_world_size_8_one_collective(tmp_path, n):
    from effocr_amd.dist import shard_sizes
    from oracle import knn_ref
    world, port = 8, _free_port()
    if n == 1024:
        assert shard_sizes(n, world) == [128] * 8
    mp.spawn(_worker8, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(1)
    X = rng.standard_normal((500, 32)).astype(np.float32)
    X[7] = X[3]
    Q = rng.standard_normal((n, 32)).astype(np.float32)
    d_ref, i_ref = knn_ref.flat_ip_search(Q, X, 10)
    for r in (0, world - 1):
        assert np.array_equal(np.load(tmp_path / f"i{r}.npy"), i_ref)
        assert np.array_equal(np.load(tmp_path / f"d{r}.npy").view(np.uint32), d_ref.view(np.uint32))


def test_pack_topk_round_trip_is_bit_exact():
    from effocr_amd.dist import pack_topk, unpack_topk
    d = torch.tensor([[1.5, float("-inf"), torch.finfo(torch.float32).min], [float("nan"), -0.0, 1e-45]])
    i = torch.tensor([[5, -1, (1 << 40) + 3], [0, -(1 << 35), 7]], dtype=torch.int64)
    buf = pack_topk(d, i)
    assert buf.dtype == torch.int32 and tuple(buf.shape) == (2, 9)
    d2, i2 = unpack_topk(buf, 3)
    assert torch.equal(d.view(torch.int32), d2.view(torch.int32)) and torch.equal(i, i2)
    with pytest.raises(ValueError):
        pack_topk(d.double(), i)


@pytest.mark.parametrize("n", [64, 33, 1])
def test_sharded_recognizer_world_size_2(tmp_path, n):
    from oracle import knn_ref
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(0)
    X = rng.standard_normal((300, 64)).astype(np.float32)
    Q = rng.standard_normal((n, 64)).astype(np.float32)
    d_ref, i_ref = knn_ref.flat_ip_search(Q, X, 5)
    for r in range(world):                                   # every rank holds the full, correctly ordered result
        assert np.array_equal(np.load(tmp_path / f"i{r}.npy"), i_ref)
        assert np.array_equal(np.load(tmp_path / f"d{r}.npy"), d_ref)


def _index_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from effocr_amd.dist import ShardedIndexSearch, shard_bounds
        from oracle import knn_ref
        rng = np.random.default_rng(1)
        X = rng.standard_normal((301, 64)).astype(np.float32)
        X[200] = X[7]; X[150] = X[7]                   # exact duplicates in BOTH shards: ties must rank by ascending global id
        Q = np.concatenate([X[[7, 250]], rng.standard_normal((19, 64)).astype(np.float32)])
        lo, hi = shard_bounds(301, rank, world)

        def search(q, k):                              # stands in for this rank's IndexFlatIP.search_device over its rows
            d, i = knn_ref.flat_ip_search(q.numpy(), X[lo:hi], k)
            return torch.from_numpy(d), torch.from_numpy(i)

        for k in (1, 10):
            d, i = ShardedIndexSearch(search, lo)(torch.from_numpy(Q), k)
            np.save(os.path.join(out_dir, f"si{rank}_{k}.npy"), i.numpy())
            np.save(os.path.join(out_dir, f"sd{rank}_{k}.npy"), d.numpy())
    finally:
        dist.destroy_process_group()


def test_sharded_index_world_size_2(tmp_path):
    """SURVEY 8(e) optional variant: index rows sharded over the ranks, per-shard top-k lists all-gathered and merged exactly — ids and
    scores bit-identical to the single-index search, duplicates across the shard boundary ranked by ascending global id."""
    from oracle import knn_ref
    world, port = 2, _free_port()
    mp.spawn(_index_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(1)
    X = rng.standard_normal((301, 64)).astype(np.float32)
    X[200] = X[7]; X[150] = X[7]
    Q = np.concatenate([X[[7, 250]], rng.standard_normal((19, 64)).astype(np.float32)])
    for k in (1, 10):
        d_ref, i_ref = knn_ref.flat_ip_search(Q, X, k)
        for r in range(world):
            assert np.array_equal(np.load(tmp_path / f"si{r}_{k}.npy"), i_ref)
            assert np.array_equal(np.load(tmp_path / f"sd{r}_{k}.npy").view(np.uint32), d_ref.view(np.uint32))
    assert list(knn_ref.flat_ip_search(Q, X, 10)[1][0][:3]) == [7, 150, 200]


def test_merge_topk_padding_and_fewer_candidates_than_k():
    from effocr_amd.dist import merge_topk
    fmin = torch.finfo(torch.float32).min
    d = torch.tensor([[[0.5, fmin]], [[0.5, 0.25]]])                 # shard 0 holds one row, shard 1 two; k' = 2
    i = torch.tensor([[[3, -1]], [[1, 9]]])
    dd, ii = merge_topk(d, i, 4)
    assert ii.tolist() == [[1, 3, 9, -1]] and dd[0, 3].item() == fmin and dd[0, 0].item() == 0.5


def test_single_process_passthrough():
    from effocr_amd.dist import ShardedRecognizer, all_gather_rows
    t = torch.arange(12).reshape(4, 3)
    assert all_gather_rows(t, 4) is t
    d, i = ShardedRecognizer(lambda x: (x.float(), x))(t)
    assert torch.equal(i, t)


def _pipeline_worker(rank, world, port, n_lines, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        from effocr_amd.dist import ShardedPipeline
        lines = [np.full((4, 8, 3), i, np.uint8) if i % 3 else f"/data/line_{i}.png" for i in range(n_lines)]
        calls = []

        def run_fn(images):                        # stands in for run_effocr on this rank's GPU: text = f(pixel) / f(path)
            calls.append(len(images))
            res = {}
            for j, im in enumerate(images):
                res[im if isinstance(im, str) else j] = f"<{im}>" if isinstance(im, str) else f"text{int(im[0, 0, 0])}"
            return res, None

        merged = ShardedPipeline(run_fn)(lines)
        assert sum(calls) <= (n_lines + world - 1) // world
        with open(os.path.join(out_dir, f"m{rank}.json"), "w") as f:
            json.dump({str(k): v for k, v in merged.items()}, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_lines", [8, 5, 1])
def test_sharded_pipeline_world_size_2(tmp_path, n_lines):
    """BASELINE configs[4], N > 1: line images sharded over ranks, ONE all-gather of the transcriptions; every rank ends with
    every line's text under the right key (paths for paths, global positions for in-memory arrays), ragged and 1-line cases."""
    import json
    world, port = 2, _free_port()
    mp.spawn(_pipeline_worker, args=(world, port, n_lines, str(tmp_path)), nprocs=world, join=True)
    want = {}
    for i in range(n_lines):
        if i % 3:
            want[str(i)] = f"text{i}"
        else:
            want[f"/data/line_{i}.png"] = f"</data/line_{i}.png>"
    for r in range(world):
        assert json.load(open(tmp_path / f"m{r}.json")) == want


def test_sharded_pipeline_single_process():
    from effocr_amd.dist import ShardedPipeline
    out = ShardedPipeline(lambda ims: ({j: f"t{int(im[0, 0, 0])}" for j, im in enumerate(ims)}, None))([np.full((1, 1, 3), 7, np.uint8)])
    assert out == {0: "t7"}
