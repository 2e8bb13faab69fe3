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
    with pytest.raises(NotImplementedError):
        LinePostprocessor(lang="en", spell_check=True)


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
