"""-m "not gpu": host-side restatements of the reference's glue (batching, id->char, blacklist,
checkpoint I/O, sharding arithmetic) and the loud no-GPU / no-extension behaviour."""
import os

import numpy as np
import pytest
import torch

from effocr_amd import pipeline, weights
from effocr_amd.dist import shard_bounds, shard_sizes

G = os.path.join(os.path.dirname(__file__), "golden")


def test_create_batches_matches_reference_convention():
    g = np.load(os.path.join(G, "pipeline.npz"))
    n, none_at = int(g["n_crops"]), int(g["none_at"])
    crops = [torch.full((3, 224, 224), float(i + 1)) for i in range(n)]
    crops[none_at] = None
    batches = pipeline.create_batches(crops)
    assert len(batches) == int(g["n_batches"]) and all(b.shape == (64, 3, 224, 224) and b.dtype == np.float32 for b in batches)
    assert (batches[0][none_at] == 0).all()                           # None -> zero image (:149-152)
    assert batches[0][0, 0, 0, 0] == 1 and batches[1][5, 0, 0, 0] == 70
    assert (batches[1][n - 64:] == 0).all() and int(g["pad_rows"]) == 64 - (n - 64)   # zero padding to 64 (:157)
    assert len(pipeline.create_batches([torch.zeros(3, 224, 224)] * 128)) == 2        # exact multiples: no pad batch


def test_iteration_returns_output_twice():
    class Eng:
        def run(self, x):
            return [x + 1]
    out = pipeline.iteration(Eng(), np.zeros((2, 3)))
    assert isinstance(out, tuple) and out[0] is out[1] and out[0][0].shape == (2, 3)   # consumers read [0][0] (:371)


def test_candidate_chars_roundtrip_and_whitespace_split(tmp_path):
    chars = ["a", "B", "日", "本", "ー"]
    p = tmp_path / "ref.txt"
    pipeline.write_candidate_chars(chars, p)
    assert p.read_text() == "a\nB\n日\n本\nー"                          # train_effocr_recognizer.py:61-62
    assert pipeline.read_candidate_chars(p) == chars                  # infer_effocr.py:203-205


def test_blacklist_filters_index_and_chars_in_the_same_order():
    g = np.load(os.path.join(G, "pipeline.npz"))
    chars = [str(c) for c in g["chars"]]

    class FakeIndex:
        def __init__(self):
            self.removed = None
        def remove_ids(self, ids):
            self.removed = np.asarray(ids)
    class FakeKnn:
        index = FakeIndex()
    knn = FakeKnn()
    kept = pipeline.apply_blacklist(knn, chars, str(g["blacklist"]))
    assert list(knn.index.removed) == [3, 7]
    assert kept == [chars[i] for i in g["kept"]]
    assert pipeline.apply_blacklist(knn, chars, None) is chars
    with pytest.raises(KeyError):
        pipeline.apply_blacklist(knn, chars, "Z")                     # dict lookup of the reference (:210)
    # golden: after compaction row ids index the filtered list
    I = g["I_after"]
    assert [kept[i] for i in I[:, 0]] == [kept[j] for j in (I[0, 0], I[1, 0], 0, len(kept) - 1)]


def test_indices_to_chars_follows_infer_effocr():
    chars = list("abcdefghij")
    idx = torch.tensor([[1, 2, 3], [9, 0, 4]])
    nearest, nns, out = pipeline.indices_to_chars(idx, chars)
    assert nearest == [["b", "c", "d"], ["j", "a", "e"]] and nns == ["bcd", "jae"] and out == "bj"
    nearest, nns, out = pipeline.indices_to_chars(torch.tensor([[4], [5]]), chars)       # k=1 works here
    assert out == "ef"
    # k > ntotal pads id -1, which the reference silently maps to candidate_chars[-1] (SURVEY a-7)
    assert pipeline.indices_to_chars(torch.tensor([[0, -1]]), chars)[0] == [["a", "j"]]


def test_checkpoint_io_and_arch_inference(tmp_path):
    sd = weights.init_state_dict("vit_tiny_test", seed=3, img_size=64)
    for name in ("enc_best.pth", "enc_best.safetensors"):
        p = tmp_path / name
        weights.save_checkpoint(sd, p)
        if name.endswith(".pth"):
            raw = torch.load(p, map_location="cpu", weights_only=True)
            assert all(k.startswith("net.") for k in raw)             # models/encoders.py:60
        back = weights.load_checkpoint(p)
        assert sorted(back) == sorted(sd) and all(torch.equal(back[k], sd[k]) for k in sd)   # safetensors sorts keys
    assert weights.infer_arch({"net." + k: v for k, v in sd.items()}) == "vit_tiny_test"
    assert weights.infer_arch(weights.init_state_dict("resnet18", 0)) == "resnet18"
    bad = dict(sd)
    bad["norm.weight"] = torch.zeros(7)
    with pytest.raises(ValueError):
        weights.check_state_dict("vit_tiny_test", bad, 64)
    with pytest.raises(NotImplementedError):
        weights.embed_dim("xcit_small_12_p8_224")


def test_seeded_init_is_deterministic():
    a = weights.init_state_dict("resnet18", seed=7)
    b = weights.init_state_dict("resnet18", seed=7)
    c = weights.init_state_dict("resnet18", seed=8)
    assert all(torch.equal(a[k], b[k]) for k in a) and not torch.equal(a["conv1.weight"], c["conv1.weight"])
    t = weights.init_state_dict("vit_small_patch16_224", seed=0, scale="timm")
    assert torch.equal(t["norm.weight"], torch.ones(384)) and t["pos_embed"].abs().max() <= 0.04


@pytest.mark.parametrize("n,world", [(1024, 8), (1000, 8), (7, 8), (0, 4), (197, 2), (5, 1)])
def test_shard_bounds_partition_exactly(n, world):
    spans = [shard_bounds(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(spans[r][1] == spans[r + 1][0] for r in range(world - 1))
    sizes = shard_sizes(n, world)
    assert sum(sizes) == n and max(sizes) - min(sizes) <= 1


def test_no_gpu_means_loud_failure_not_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from effocr_amd import EffOCRHipError
    from effocr_amd.encoders import AutoEncoderFactory, HipEncoder
    from effocr_amd.knn import IndexFlatIP
    sd = weights.init_state_dict("vit_tiny_test", seed=0, img_size=64)
    with pytest.raises(EffOCRHipError):
        HipEncoder("vit_tiny_test", sd, img_size=64)
    with pytest.raises(EffOCRHipError):
        IndexFlatIP(128)
    enc = AutoEncoderFactory("timm", "vit_tiny_test", img_size=64)()
    enc.to("cuda").eval()
    assert sum(p.numel() for _, p in enc.named_parameters()) > 0       # infer_effocr.py:538-540 works without a GPU
    with pytest.raises(EffOCRHipError):
        enc(torch.zeros(1, 3, 64, 64))
    with pytest.raises(NotImplementedError):
        AutoEncoderFactory("hf", "facebook/dino-vits16")
    with pytest.raises(NotImplementedError):
        AutoEncoderFactory("timm", "xcit_small_12_p8_224")


def test_product_package_never_imports_the_oracle():
    import pathlib
    root = pathlib.Path(__file__).resolve().parents[1] / "effocr_amd"
    for p in root.rglob("*.py"):
        src = p.read_text()
        assert "import oracle" not in src and "from oracle" not in src, p


def test_transform_box_resolution_matches_numpy_slicing():
    """effocr_amd.transforms.slice_boxes / round_boxes (host side of the device crop transform)."""
    from effocr_amd.transforms import round_boxes, slice_boxes
    im = np.zeros((30, 50, 3), np.uint8)
    boxes = [(1, 2, 45, 29), (-10, 0, 50, 30), (0, -5, 80, 90), (49, 29, 50, 30)]
    ib = slice_boxes(boxes, 30, 50)
    for (x0, y0, x1, y1), b in zip(ib, boxes):
        assert im[y0:y1, x0:x1].shape == im[b[1]:b[3], b[0]:b[2]].shape
    assert ib.dtype == np.int32
    for bad in [(5, 5, 5, 9), (9, 5, 3, 9), (60, 0, 70, 10), (0, 0, 10, -40)]:
        with pytest.raises(ValueError):
            slice_boxes([bad], 30, 50)
    assert round_boxes([(0.5, 1.5, 2.5, 3.49, 0.9)]) == [(0, 2, 2, 3)]


# ---------------------------------------------------------------- line pre/post-processing
def _boxes(specs):
    return np.array(specs, dtype=np.float32)


def test_en_preprocess_word_starts_and_quirks():
    from effocr_amd.postprocess import LinePostprocessor
    from oracle import postprocess_ref as R
    #            x0  y0  x1  y1  score           characters given out of order, one below threshold
    chars = _boxes([[30, 2, 38, 20, .9], [0, 2, 8, 20, .9], [10, 5, 18, 20, .9], [20, 3, 28, 20, .4], [44, 2, 52, 20, .8], [54, 2, 60, 20, .7]])
    words = _boxes([[43, 0, 61, 22, .9], [0, 0, 39, 22, .95], [100, 0, 120, 22, .9], [5, 0, 9, 9, .2]])
    post = LinePostprocessor(lang="en")
    got_c, got_w = post.en_preprocess((chars, words))
    ref_c, ref_w = R.en_preprocess(chars, words)
    assert [list(map(float, c)) for c in got_c] == [list(map(float, c)) for c in ref_c]
    assert [c[0] for c in got_c] == [0, 10, 30, 44, 54]               # sorted by x0, score 0.4 dropped
    # word at x=0 starts at char 0, word at x=43 at char 3; the word at x=100 has no candidate and REPEATS 3
    assert got_w == ref_w == [0, 3, 3]
    # mmdet-style nesting: result[0] = (chars, words)
    assert post.en_preprocess([(chars, words)])[1] == got_w
    # vertical: sort by y0
    v = LinePostprocessor(lang="en", vertical=True)
    assert [c[1] for c in v.en_preprocess((chars, words))[0]] == sorted(c[1] for c in ref_c)
    jp = LinePostprocessor(lang="jp", score_thresh=0.75)
    assert [c[0] for c in jp.jp_preprocess([[chars]])] == [0, 10, 30, 44]


def test_en_postprocess_spaces_case_and_period():
    from effocr_amd.postprocess import LinePostprocessor
    from oracle import postprocess_ref as R
    line = "theCATran-"
    #        t   h   e   C   A   T   r   a   n   -
    hts = [18, 18, 10, 10, 10, 18, 10, 10, 10, 3]
    bots = [20, 20, 20, 20, 20, 20, 20, 20, 20, 19.5]
    wei = [0, 3, 6]
    for margin in (None, 0.15):
        post = LinePostprocessor(lang="en", anchor_margin=margin)
        got = post.en_postprocess(line, wei, hts, bots)
        assert got == R.en_postprocess(line, wei, hts, bots, anchor_margin=margin)
    assert LinePostprocessor(lang="en").en_postprocess(line, wei, hts, bots) == "the CAT ran-"
    # anchors e, a, n, r (height 10): chars within 15 % are lowered (C, A -> c, a); '-' on the baseline -> '.'
    assert LinePostprocessor(lang="en", anchor_margin=0.15).en_postprocess(line, wei, hts, bots) == "the caT ran."
    # raising: a nondistinct lowercase letter far taller than the anchors
    got = LinePostprocessor(lang="en", anchor_margin=0.1).en_postprocess("anso", [0], [10, 10, 20, 10], [5, 5, 5, 5])
    assert got == R.en_postprocess("anso", [0], [10, 10, 20, 10], [5, 5, 5, 5], anchor_margin=0.1) == "anSo"
    # quirks: no word boxes -> None; length mismatch -> AssertionError
    assert LinePostprocessor(lang="en").en_postprocess("abc", [], [1, 1, 1], [1, 1, 1]) is None
    assert R.en_postprocess("abc", [], [1, 1, 1], [1, 1, 1]) is None
    with pytest.raises(AssertionError):
        LinePostprocessor(lang="en").en_postprocess("abc", [0], [1, 1], [1, 1, 1])
    with pytest.raises(RuntimeError):                      # spell_check=True needs a word-frequency dictionary (symspellpy absent here)
        LinePostprocessor(lang="en", spell_check=True)
    assert LinePostprocessor(lang="en", spell_check=True, worddict={"hello": 5}).en_postprocess("he11o", [0], [9.0] * 5, [1.0] * 5) == "hello"


def test_postprocess_matches_restatement_on_random_lines():
    from effocr_amd.postprocess import LinePostprocessor
    from oracle import postprocess_ref as R
    rng = np.random.default_rng(0)
    alphabet = list("aenrwuosvcxzTHKQ-.")
    for trial in range(200):
        n = int(rng.integers(1, 30))
        line = "".join(rng.choice(alphabet, n))
        hts = [float(h) for h in rng.choice([3, 9, 10, 11, 18, 19, 60], n)]
        bots = [float(b) for b in rng.choice([19.0, 20.0, 20.5, 25.0], n)]
        wei = sorted(set(int(i) for i in rng.integers(0, n, int(rng.integers(1, 5)))))
        margin = [None, 0.1, 0.25][trial % 3]
        post = LinePostprocessor(lang="en", anchor_margin=margin)
        assert post.en_postprocess(line, wei, hts, bots) == R.en_postprocess(line, wei, hts, bots, anchor_margin=margin), (line, wei)


def test_ref_index_byte_exact_fixture_and_rejects(tmp_path):
    """train_effocr_recognizer.py:52 writes ``ref.index`` with faiss.write_index(IndexFlatIP).  Byte layout restated from
    faiss index_write.cpp (write_index_header + IndexFlat): fourcc "IxFI" | d i32 | ntotal i64 | dummy i64 (1<<20) |
    dummy i64 (1<<20) | is_trained u8 | metric_type i32 (0 = METRIC_INNER_PRODUCT) | vector size u64 | floats.  The fixture
    below is assembled BY HAND, byte for byte, independently of effocr_amd.knn.write_index."""
    import struct
    from unittest import mock
    import effocr_amd.knn as K
    rows = np.array([[1.0, -2.5, 0.0], [0.5, 0.25, -1.0]], dtype="<f4")
    blob = (b"IxFI" + bytes([3, 0, 0, 0]) + bytes([2, 0, 0, 0, 0, 0, 0, 0]) + bytes([0, 0, 0x10, 0, 0, 0, 0, 0]) * 2 +
            bytes([1]) + bytes([0, 0, 0, 0]) + bytes([6, 0, 0, 0, 0, 0, 0, 0]) + rows.tobytes())
    assert len(blob) == 4 + 4 + 8 + 8 + 8 + 1 + 4 + 8 + 24
    path = tmp_path / "ref.index"
    path.write_bytes(blob)

    class FakeIndex:                                   # the parser itself needs no GPU: capture what it would upload
        def __init__(self, d, device=None):
            self.d, self.added = d, None

        def add(self, x):
            self.added = np.array(x)

    with mock.patch.object(K, "IndexFlatIP", FakeIndex):
        idx = K.read_index(str(path))
    assert idx.d == 3 and np.array_equal(idx.added, rows)

    class FakeOut:
        d, ntotal = 3, 2
        _xb = torch.from_numpy(rows.astype(np.float32))

    out = tmp_path / "out.index"
    K.write_index(FakeOut, str(out))
    assert out.read_bytes() == blob                    # the writer reproduces the hand-assembled file byte for byte

    def rejected(b):
        p = tmp_path / "bad.index"
        p.write_bytes(b)
        with mock.patch.object(K, "IndexFlatIP", FakeIndex), pytest.raises(ValueError):
            K.read_index(str(p))

    rejected(b"IxF2" + blob[4:])                       # IndexFlatL2
    rejected(b"IxFl" + blob[4:])                       # legacy IndexFlat fourcc
    rejected(b"IwFl" + blob[4:])                       # IVF
    rejected(blob[:-4])                                # truncated payload
    rejected(blob[:20])                                # truncated header
    rejected(blob[:33] + bytes([1, 0, 0, 0]) + blob[37:])          # metric_type 1 (L2) under the IP fourcc
    rejected(blob[:37] + bytes([7, 0, 0, 0, 0, 0, 0, 0]) + blob[45:])   # vector size != ntotal * d


def test_checkpoint_variants_of_the_reference_writer(tmp_path):
    """train_effocr_recognizer.py:65-72 saves ``encoder.state_dict()`` of a timm model wrapped as ``self.net``: BatchNorm
    buffers include ``num_batches_tracked`` (int64 scalars); Lightning-style files wrap it in {"state_dict": ...}."""
    from effocr_amd import weights as W
    sd = W.init_state_dict("resnet18", seed=3)
    ref_style = {"net." + k: v for k, v in sd.items()}
    for k in list(sd):
        if k.endswith("running_var"):
            ref_style["net." + k.replace("running_var", "num_batches_tracked")] = torch.tensor(1234, dtype=torch.int64)
    p1 = tmp_path / "enc_best.pth"
    torch.save(ref_style, p1)
    got = W.load_checkpoint(p1)
    W.check_state_dict("resnet18", got, 32)            # extra buffers are ignored, nothing missing
    assert W.infer_arch(got) == "resnet18" and all(torch.equal(got[k], sd[k]) for k in sd)
    p2 = tmp_path / "wrapped.pth"
    torch.save({"state_dict": ref_style, "epoch": 3}, p2)
    got2 = W.load_checkpoint(p2)
    assert all(torch.equal(got2[k], sd[k]) for k in sd)
    bad = dict(ref_style)
    del bad["net.conv1.weight"]
    p3 = tmp_path / "bad.pth"
    torch.save(bad, p3)
    with pytest.raises(ValueError):
        W.check_state_dict("resnet18", W.load_checkpoint(p3), 32)


def test_executor_thread_drains_queue_and_keeps_order():
    """infer_effocr_onnx_multi.py:207-223,350-369 with a fake engine: every batch is run exactly once, results come back
    indexed by batch number whatever thread ran them, and an engine error surfaces instead of hanging."""
    import queue
    from effocr_amd.pipeline import RecognizerEngineExecutorThread

    class Engine:
        def __init__(self):
            self.calls = []

        def run(self, b):
            self.calls.append(int(b[0]))
            if b[0] < 0:
                raise RuntimeError("boom")
            return [b * 2]

    eng, qi, qo = Engine(), queue.Queue(), queue.Queue()
    for i in range(11):
        qi.put((i, np.array([i])))
    ths = [RecognizerEngineExecutorThread(eng, qi, qo) for _ in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    res = {}
    while not qo.empty():
        i, out = qo.get()
        res[i] = out
    assert sorted(res) == list(range(11)) and sorted(eng.calls) == list(range(11))
    assert all(res[i][0][0][0] == 2 * i and res[i][0] is res[i][1] for i in res)      # (output, output); [0][0] = embeddings
    qi.put((0, np.array([-1])))
    t = RecognizerEngineExecutorThread(eng, qi, qo)
    t.start()
    t.join()
    assert isinstance(t.error, RuntimeError)


def test_engines_default_to_the_reference_conventions():
    """double_clipped is hard-coded True in the reference (infer_effocr.py:226); devices follow the encoder."""
    import inspect
    from effocr_amd.knn import InferenceModel
    from effocr_amd.pipeline import Recognizer, encoder_device
    from effocr_amd.postprocess import LineRecognizer
    assert inspect.signature(LineRecognizer.__init__).parameters["double_clipped"].default is True
    assert inspect.signature(Recognizer.recognize_boxes).parameters["double_clipped"].default is True
    # round 4: the default operand precision is the one that meets north_star's 1e-3 tolerance
    from effocr_amd.encoders import DEFAULT_PRECISION, AutoEncoderFactory, HipEncoder
    from effocr_amd.recognizer_engine import EffRecognizer
    assert DEFAULT_PRECISION == "fp16"
    for fn in (HipEncoder.__init__, AutoEncoderFactory, EffRecognizer.__init__):
        assert inspect.signature(fn).parameters["precision"].default == "fp16"
    assert inspect.signature(Recognizer.recognize_boxes).parameters["vertical"].default is False

    class Trunk:
        _device = torch.device("cuda:3")

    assert encoder_device(Trunk()) == torch.device("cuda:3")
    assert InferenceModel(Trunk(), knn_func=object()).data_device == torch.device("cuda:3")
    assert InferenceModel(Trunk(), knn_func=object(), data_device="cuda:1").data_device == torch.device("cuda:1")


def test_bench_self_launch_and_scaling_flags():
    """bench.py contract: plain ``python bench.py --gpus N`` must not die before becoming the launcher; the N>1 default is the
    BASELINE configs[2] (strong) split; the sharding helper gives 128 crops per rank at 1024 / 8."""
    import importlib.util
    import os
    import sys
    from effocr_amd.dist import shard_bounds
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    try:
        sys.argv = ["bench.py", "--gpus", "8"]
        a = mod.parse()
    finally:
        sys.argv = argv
    assert a.scaling == "auto" and a.gpus == 8 and a.batch == 1024
    assert [shard_bounds(1024, r, 8) for r in (0, 7)] == [(0, 128), (896, 1024)]
    src = open(spec.origin).read()
    assert "torch.distributed.run" in src and "subprocess.call" in src


def test_yolov5s_table_and_letterbox_geometry():
    """The layer table has ultralytics' published YOLOv5s (v6) size — 7,235,389 parameters at 80 classes ("YOLOv5s summary:
    270 layers, 7235389 parameters") — and letterbox_geometry is the arithmetic of localizer_engine.py:107-138."""
    from effocr_amd.localizer_engine import init_yolov5s_state_dict, letterbox_geometry, yolov5s_param_shapes
    from oracle import yolo_ref as Y
    sh = yolov5s_param_shapes(80)
    n = sum(int(np.prod(v)) for k, v in sh.items() if not k.endswith(("running_mean", "running_var", "anchors")))
    assert n == 7235389
    sd = init_yolov5s_state_dict(2, seed=0)
    assert set(sd) == set(yolov5s_param_shapes(2)) and all(tuple(sd[k].shape) == tuple(v) for k, v in yolov5s_param_shapes(2).items())
    assert torch.equal(sd["model.0.conv.weight"], init_yolov5s_state_dict(2, seed=0)["model.0.conv.weight"])
    for hw in [(50, 300), (256, 4096), (640, 640), (1000, 37), (481, 640)]:
        nh, nw, top, bottom, left, right, ratio, _ = letterbox_geometry(hw, (640, 640), auto=False)
        im = np.zeros(hw + (3,), dtype=np.uint8)
        out, r2, _ = Y.letterbox(im, (640, 640), auto=False)
        assert out.shape[:2] == (nh + top + bottom, nw + left + right) == (640, 640) and r2 == ratio
    # hand-computed: 50 x 300 -> r = min(12.8, 2.1333) -> 640 x 107, dh = 533 / 2 = 266.5 -> top 266, bottom 267
    assert letterbox_geometry((50, 300), (640, 640), auto=False)[:6] == (107, 640, 266, 267, 0, 0)


def test_yolo_oracle_nms_and_resize_hand_cases():
    """The restated pieces on cases small enough to check by hand."""
    from oracle import yolo_ref as Y
    # three boxes of one class: B overlaps A (IoU 0.6) and is less confident -> dropped at 0.5, kept at 0.7; C is elsewhere
    pred = torch.tensor([[[50, 50, 20, 20, 0.9, 0.9, 0.1], [52.5, 50, 20, 20, 0.8, 0.9, 0.1], [200, 50, 20, 20, 0.7, 0.2, 0.9]]])
    o = Y.non_max_suppression(pred, 0.25, 0.5, max_det=10)[0]
    assert o.shape == (2, 6) and o[0, :4].tolist() == [40, 40, 60, 60] and abs(o[0, 4] - 0.81) < 1e-6 and o[1, 5] == 1
    assert Y.non_max_suppression(pred, 0.25, 0.8, max_det=10)[0].shape == (3, 6)
    # same box, different classes: the class offset keeps both
    pred2 = torch.tensor([[[50, 50, 20, 20, 0.9, 0.9, 0.1], [50, 50, 20, 20, 0.8, 0.1, 0.9]]])
    assert Y.non_max_suppression(pred2, 0.25, 0.1, max_det=10)[0].shape == (2, 6)
    assert Y.non_max_suppression(pred2, 0.25, 0.1, agnostic=True, max_det=10)[0].shape == (1, 6)
    # objectness passes but obj * cls does not
    assert Y.non_max_suppression(torch.tensor([[[5, 5, 2, 2, 0.5, 0.5, 0.5]]]), 0.3, 0.5)[0].shape == (0, 6)
    # 2x upscale of a 1 x 2 image: half-pixel centres -> [a, (3a+b)/4, (a+3b)/4, b] in cv2's fixed point
    im = np.array([[[0, 0, 0], [200, 100, 40]]], dtype=np.uint8)
    r = Y.resize_linear_u8(im, 4, 1)
    assert r[0, :, 0].tolist() == [0, 50, 150, 200] and r[0, :, 1].tolist() == [0, 25, 75, 100]
    assert np.array_equal(Y.resize_linear_u8(im, 2, 1), im)
