"""-m gpu: the engine objects under the reference's threading / multi-process call patterns, and the precision
modes' top-1 safety margin (north_star: identical top-1 ids)."""
import os
import socket
import threading

import numpy as np
import pytest
import torch

from effocr_amd.weights import init_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("staging,lanes", [("direct", 4), ("direct", 2), ("pinned", 2)])
def test_effrecognizer_shared_by_threads(dev, staging, lanes):
    """infer_effocr_onnx_multi.py:207-223,350-364: N Python threads call ``run`` on ONE engine instance.  Results must be
    bit-identical to serial calls (per-call streams, one workspace per stream; the host batch reaches the device by the runtime's
    pageable copy — the default — or through the pinned staging slices)."""
    from effocr_amd.recognizer_engine import EffRecognizer
    arch = "vit_tiny_test"
    sd = init_state_dict(arch, seed=4, img_size=64)
    eng = EffRecognizer(sd, arch=arch, precision="bf16", img_size=64, device=dev, lanes=lanes, staging=staging)
    rng = np.random.default_rng(1)
    batches = [rng.standard_normal((b, 3, 64, 64), dtype=np.float32) for b in (64, 64, 7, 64, 33, 64, 1, 64)]
    serial = [eng.run(b)[0] for b in batches]
    results = [[None] * len(batches) for _ in range(4)]
    errors = []

    def worker(t):
        try:
            for rep in range(3):
                for i, b in enumerate(batches):
                    out = eng(b)
                    assert isinstance(out, list) and out[0].dtype == np.float32 and out[0].shape == (b.shape[0], 128)
                    results[t][i] = out[0]
        except Exception as e:                 # pragma: no cover
            errors.append(e)

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    for t in range(4):
        for i in range(len(batches)):
            assert np.array_equal(results[t][i], serial[i]), (t, i)
    assert eng.run(np.zeros((0, 3, 64, 64), dtype=np.float32))[0].shape == (0, 128)


def test_executor_threads_match_serial(dev):
    """run_recognizer_batches with the reference's executor threads (num_streams=4) == the serial loop."""
    from effocr_amd.knn import FaissKNN, IndexFlatIP
    from effocr_amd.pipeline import run_recognizer_batches
    from effocr_amd.recognizer_engine import EffRecognizer
    arch = "vit_tiny_test"
    sd = init_state_dict(arch, seed=5, img_size=224)
    eng = EffRecognizer(sd, arch=arch, precision="fp32", img_size=224, device=dev)
    g = torch.Generator().manual_seed(2)
    crops = [torch.randn(3, 224, 224, generator=g) for _ in range(150)]
    crops[17] = None
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False, device=dev)
    knn.train(torch.nn.functional.normalize(torch.randn(97, 128, generator=g), dim=1))
    chars = [chr(0x4E00 + i) for i in range(97)]
    c1, f1 = run_recognizer_batches(crops, eng, knn, chars, num_streams=1)
    c4, f4 = run_recognizer_batches(crops, eng, knn, chars, num_streams=4)
    assert f1 == f4 and c1 == c4 and len(f1) == 192          # 3 batches of 64 (the padded tail included, as in the reference)


def test_streams_do_not_share_workspaces(dev):
    """Two HIP streams forwarding through one encoder / one index concurrently get the serial results."""
    from effocr_amd.encoders import HipEncoder
    from effocr_amd.knn import IndexFlatIP
    arch = "vit_tiny_test"
    enc = HipEncoder(arch, init_state_dict(arch, seed=6, img_size=64), img_size=64, precision="bf16", device=dev)
    idx = IndexFlatIP(128, device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    idx.add(torch.nn.functional.normalize(torch.randn(5000, 128, generator=g, device=dev), dim=1))
    xa = torch.randn(200, 3, 64, 64, generator=g, device=dev)
    xb = torch.randn(300, 3, 64, 64, generator=g, device=dev)
    ea, eb = enc.forward(xa, normalize=True), enc.forward(xb, normalize=True)
    ia, ib = idx.search_device(ea, 10)[1], idx.search_device(eb, 10)[1]
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for _ in range(5):
        with torch.cuda.stream(sa):
            ra = idx.search_device(enc.forward(xa, normalize=True), 10)[1]
        with torch.cuda.stream(sb):
            rb = idx.search_device(enc.forward(xb, normalize=True), 10)[1]
        torch.cuda.synchronize()
        assert torch.equal(ra, ia) and torch.equal(rb, ib)
    assert len(enc._ws) >= 3 and len(idx._ws) >= 3          # default stream + two side streams


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)       # both ranks share the one GPU of the test box: RCCL refuses that
    try:
        from effocr_amd.dist import ShardedRecognizer
        from effocr_amd.encoders import AutoEncoderFactory
        from effocr_amd.knn import FaissKNN, IndexFlatIP
        from effocr_amd.pipeline import Recognizer
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(rank % ndev)
        dev = torch.device("cuda", rank % ndev)
        arch = "vit_tiny_test"
        enc = AutoEncoderFactory("timm", arch, precision="fp32", img_size=64)()
        enc.load_state_dict(init_state_dict(arch, seed=8, img_size=64))
        enc.to(dev).eval()
        g = torch.Generator().manual_seed(9)
        index = torch.nn.functional.normalize(torch.randn(300, 128, generator=g), dim=1)
        knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False)    # device = the current device
        knn.train(index)
        rec = Recognizer(enc, knn, [chr(0x3041 + i % 80) for i in range(300)], knn=10)
        x = torch.randn(37, 3, 64, 64, generator=g)
        d, i = ShardedRecognizer(rec.neighbors)(x)
        assert str(knn.index.device) == str(dev) and str(rec.recognizer.data_device) == str(dev)
        torch.save((d.cpu(), i.cpu()), os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_sharded_recognizer_two_processes_on_gpu(dev, tmp_path):
    """SURVEY 8e through the product API: two processes (gloo rendezvous; one process per GPU when the box has two,
    otherwise both on the one GPU), each running the HIP Recognizer on its shard, ids all-gathered == single process."""
    import torch.multiprocessing as mp
    from effocr_amd.encoders import AutoEncoderFactory
    from effocr_amd.knn import FaissKNN, IndexFlatIP
    from effocr_amd.pipeline import Recognizer
    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    d0, i0 = torch.load(tmp_path / "r0.pt")
    d1, i1 = torch.load(tmp_path / "r1.pt")
    assert torch.equal(i0, i1) and torch.equal(d0, d1) and tuple(i0.shape) == (37, 10)
    arch = "vit_tiny_test"
    enc = AutoEncoderFactory("timm", arch, precision="fp32", img_size=64)()
    enc.load_state_dict(init_state_dict(arch, seed=8, img_size=64))
    enc.to(dev).eval()
    g = torch.Generator().manual_seed(9)
    index = torch.nn.functional.normalize(torch.randn(300, 128, generator=g), dim=1)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False)
    knn.train(index)
    rec = Recognizer(enc, knn, ["x"] * 300, knn=10)
    x = torch.randn(37, 3, 64, 64, generator=g)
    d, i = rec.neighbors(x)
    assert torch.equal(i.cpu(), i0)
    assert torch.equal(d.cpu(), d0)                     # same kernels, same rows: bit-identical however the batch is split


def _index_shard_main(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)       # (both ranks may share the test box's one GPU)
    try:
        from effocr_amd.dist import ShardedIndexSearch, shard_bounds
        from effocr_amd.knn import IndexFlatIP
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(rank % ndev)
        dev = torch.device("cuda", rank % ndev)
        X, Q = _index_shard_data()
        lo, hi = shard_bounds(X.shape[0], rank, world)
        idx = IndexFlatIP(X.shape[1], device=dev, screen=False)
        idx.add(X[lo:hi])
        for k in (1, 10):
            d, i = ShardedIndexSearch(idx.search_device, lo)(Q.to(dev), k)
            torch.save((d.cpu(), i.cpu()), os.path.join(out_dir, f"s{rank}_{k}.pt"))
    finally:
        dist.destroy_process_group()


def _index_shard_data():
    g = torch.Generator().manual_seed(31)
    X = torch.nn.functional.normalize(torch.randn(70_001, 128, generator=g), dim=1)
    X[60_000] = X[5]; X[20_000] = X[5]                                  # duplicates on both sides of the shard boundary
    Q = torch.nn.functional.normalize(torch.cat([X[[5, 69_000]], torch.randn(21, 128, generator=g)]), dim=1)
    return X, Q


def test_index_sharded_search_two_processes_on_gpu(dev, tmp_path):
    """SURVEY 8(e)'s optional variant through the product API: the index rows split over two processes (HIP streaming search per shard),
    (score, global id) lists all-gathered and merged — bit-identical to one process searching the whole index, duplicates ranked by id."""
    import torch.multiprocessing as mp
    from effocr_amd.knn import IndexFlatIP
    mp.spawn(_index_shard_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    X, Q = _index_shard_data()
    idx = IndexFlatIP(X.shape[1], device=dev, screen=False)
    idx.add(X)
    for k in (1, 10):
        d, i = idx.search_device(Q.to(dev), k)
        for r in range(2):
            dr, ir = torch.load(tmp_path / f"s{r}_{k}.pt")
            assert torch.equal(ir, i.cpu()) and torch.equal(dr.view(torch.int32), d.cpu().view(torch.int32))
    assert idx.search_device(Q.to(dev), 10)[1][0, :3].tolist() == [5, 20_000, 60_000]


def _nccl_rank_main(rank, world, port, out_dir):
    """One process per GPU over RCCL ("nccl"): the product's ShardedRecognizer + the id all-gather exactly as bench.py --gpus N runs it."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from effocr_amd.dist import ShardedRecognizer, all_gather_rows, shard_bounds
        from effocr_amd.encoders import AutoEncoderFactory
        from effocr_amd.knn import FaissKNN, IndexFlatIP
        from effocr_amd.pipeline import Recognizer
        arch = "vit_tiny_test"
        enc = AutoEncoderFactory("timm", arch, precision="fp32", img_size=64)()
        enc.load_state_dict(init_state_dict(arch, seed=8, img_size=64))
        enc.to(dev).eval()
        g = torch.Generator().manual_seed(9)
        index = torch.nn.functional.normalize(torch.randn(300, 128, generator=g), dim=1)
        knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False)
        knn.train(index)
        rec = Recognizer(enc, knn, ["x"] * 300, knn=10)
        x = torch.randn(37, 3, 64, 64, generator=g)
        d, i = ShardedRecognizer(rec.neighbors)(x)
        # the raw collective with a ragged int64 block, forced even for a one-rank group: all_gather_into_tensor over RCCL
        lo, hi = shard_bounds(37, rank, world)
        rows = torch.arange(37 * 10, dtype=torch.int64, device=dev).view(37, 10)
        full = all_gather_rows(rows[lo:hi].contiguous(), 37, always_collective=True)
        assert torch.equal(full, rows)
        torch.save((d.cpu(), i.cpu()), os.path.join(out_dir, f"n{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _single_process_reference(dev):
    from effocr_amd.encoders import AutoEncoderFactory
    from effocr_amd.knn import FaissKNN, IndexFlatIP
    from effocr_amd.pipeline import Recognizer
    arch = "vit_tiny_test"
    enc = AutoEncoderFactory("timm", arch, precision="fp32", img_size=64)()
    enc.load_state_dict(init_state_dict(arch, seed=8, img_size=64))
    enc.to(dev).eval()
    g = torch.Generator().manual_seed(9)
    index = torch.nn.functional.normalize(torch.randn(300, 128, generator=g), dim=1)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False)
    knn.train(index)
    rec = Recognizer(enc, knn, ["x"] * 300, knn=10)
    return rec.neighbors(torch.randn(37, 3, 64, 64, generator=g))


def test_rccl_all_gather_single_rank_group(dev, tmp_path):
    """The RCCL leg on the one-GPU test box: a "nccl" process group of ONE rank (its own process), the product's ShardedRecognizer and
    a forced ``all_gather_into_tensor`` of int64 id rows — so the collective call, dtype and padding logic have executed on RCCL at
    least once before the driver's multi-GPU bench does."""
    import torch.multiprocessing as mp
    mp.spawn(_nccl_rank_main, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    d0, i0 = torch.load(tmp_path / "n0.pt")
    d, i = _single_process_reference(dev)
    assert torch.equal(i.cpu(), i0) and torch.equal(d.cpu(), d0)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs a second GPU")
def test_sharded_recognizer_two_processes_over_rccl(dev, tmp_path):
    """BASELINE configs[2] in miniature: two processes, one per GPU, "nccl" = RCCL over xGMI; gathered ids / scores == single process."""
    import torch.multiprocessing as mp
    mp.spawn(_nccl_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    d0, i0 = torch.load(tmp_path / "n0.pt")
    d1, i1 = torch.load(tmp_path / "n1.pt")
    d, i = _single_process_reference(dev)
    assert torch.equal(i0, i1) and torch.equal(d0, d1)
    assert torch.equal(i.cpu(), i0) and torch.equal(d.cpu(), d0)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs a second GPU")
def test_recognizer_follows_the_encoder_device():
    """the engines derive their device from the encoder / current device, not a literal cuda:0."""
    from effocr_amd.encoders import AutoEncoderFactory
    from effocr_amd.knn import FaissKNN, IndexFlatIP
    from effocr_amd.pipeline import Recognizer
    dev1 = torch.device("cuda:1")
    torch.cuda.set_device(dev1)
    try:
        arch = "vit_tiny_test"
        enc = AutoEncoderFactory("timm", arch, precision="fp32", img_size=64)()
        enc.to(dev1).eval()
        knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False)
        knn.train(torch.nn.functional.normalize(torch.randn(50, 128), dim=1))
        rec = Recognizer(enc, knn, ["a"] * 50, knn=3)
        d, i = rec.neighbors(torch.randn(5, 3, 64, 64))
        assert i.device == dev1 and knn.index.device == dev1
    finally:
        torch.cuda.set_device(0)


def test_top1_safety_margin_per_precision(dev):
    """north_star asks for IDENTICAL top-1 ids.  A reduced-precision embedding e' can only change the top-1 between two
    index rows a, b when their scores differ by less than e'.(a-b) noise.  Worst case over directions: with the error
    component perpendicular to the exact embedding of norm r, an adversarial pair (a = e, b = rotated by angle t) flips
    iff its cosine margin 1 - cos t is below ~ r^2 / 2.  This measures r for every precision mode on ViT-S/16 (oracle A
    as the exact embedding), checks the flips on planted adversarial pairs, and asserts the safe margins:
    fp32 < 5e-7, fp16 < 2e-6, bf16 < 1e-4 — i.e. bf16 (the BASELINE configuration) can reorder only glyphs whose cosine
    scores are closer than 1e-4."""
    from effocr_amd.encoders import HipEncoder
    from effocr_amd.knn import IndexFlatIP
    from oracle.encoders_ref import encoder_forward, l2_normalize
    arch = "vit_small_patch16_224"
    sd = init_state_dict(arch, seed=0)
    x = torch.randn(48, 3, 224, 224, generator=torch.Generator().manual_seed(12))
    e = l2_normalize(encoder_forward(arch, sd, x)).double()
    bounds = {"fp32": 5e-7, "fp16": 2e-6, "bf16": 1e-4}   # fp32: limited by the fp32 resolution of the scores themselves
    report = {}
    for prec in ("fp32", "fp16", "bf16"):
        eh = HipEncoder(arch, sd, precision=prec, device=dev).forward(x.to(dev), normalize=True).cpu().double()
        err = eh - e
        perp = err - (err * e).sum(1, keepdim=True) * e
        r = perp.norm(dim=1)
        safe = float((r * r / 2).max())
        report[prec] = safe
        assert safe < bounds[prec], (prec, safe)
        # planted pairs at twice the safe margin never flip, even in the worst direction (along the error itself)
        u = perp / r[:, None]
        t = torch.sqrt(torch.tensor(2 * 2.0 * bounds[prec], dtype=torch.float64))
        b = torch.nn.functional.normalize(e + t * u, dim=1)
        idx = IndexFlatIP(384, device=dev)
        X = torch.stack([e, b], dim=1).reshape(-1, 384).float()
        idx.add(X)
        I = idx.search_device(eh.float().to(dev), 1)[1][:, 0].cpu()
        assert torch.equal(I, 2 * torch.arange(48)), prec
    assert report["fp16"] < report["bf16"]
    print("top-1 safe cosine margins:", {k: f"{v:.2e}" for k, v in report.items()})


def test_stream_split_forward_from_several_caller_threads(dev):
    """Round 6: 256-crop calls run as concurrent sub-batches on the encoder's side streams.  Three caller threads, each on a stream of
    its own, share ONE HipEncoder (the reference's N-threads-one-engine use, infer_effocr_onnx_multi.py:207-223): the side streams and
    their workspaces are shared too, so the sub-batches of different callers queue behind each other — results must stay bit-identical
    to the same calls made one after the other."""
    from effocr_amd.encoders import HipEncoder
    arch = "vit_small_patch16_224"
    enc = HipEncoder(arch, init_state_dict(arch, seed=2, img_size=224), precision="fp16", device=dev)
    g = torch.Generator(device=dev).manual_seed(9)
    xs = [torch.randn(b, 3, 224, 224, generator=g, device=dev) for b in (256, 200, 300)]
    assert all(enc._split_plan(x.shape[0]) > 1 for x in xs)
    serial = [enc.forward(x, normalize=True).clone() for x in xs]
    torch.cuda.synchronize()
    out, errors = [None] * 3, []

    def worker(i):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(8):                         # (callers' sub-batches meet on the shared side streams: one forward at a time per stream)
                    r = enc.forward(xs[i], normalize=True)
                st.synchronize()
            out[i] = r
        except Exception as e:                 # pragma: no cover
            errors.append(e)

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    for i in range(3):
        assert torch.equal(out[i], serial[i]), i
    enc.check_status()
