"""Host-logic rows (SURVEY §8 a-9, a-10 parts, f-3 static helpers, f-4) pinned to THE REFERENCE ITSELF: the fixtures under
tests/golden/ref_*.{json,npz} hold inputs and outputs of the reference's own functions, recorded by
tests/golden/make_ref_golden.py, which imports /root/reference in the build container (stubs for the absent third-party
packages, armed so that no stub value can reach a recorded output).  Both the PRODUCT functions (effocr_amd/) and the
ORACLE restatements (oracle/) are checked against them; the last test replays the reference's whole ``run_effocr``
(infer_effocr_onnx_multi.py:227-397) through oracle/run_effocr_ref.py.  Nothing here reads /root/reference."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(G, "ref_hostlogic.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def fa():
    return np.load(os.path.join(G, "ref_hostlogic.npz"))


def _floats(rows):
    return [[float(v) for v in r] for r in rows]


def test_constants_are_the_references(fx):
    from effocr_amd import pipeline, postprocess
    assert fx["LARGE_NUMBER_multi"] == 1_000_000_000 == pipeline.LARGE_NUMBER            # infer_effocr_onnx_multi.py:46
    assert "".join(fx["distinct_lowercase"]) == postprocess.DISTINCT_LOWERCASE
    assert "".join(fx["nondistinct_lowercase"]) == postprocess.NONDISTINCT_LOWERCASE
    assert fx["COCO_JSON_SKELETON"] == pipeline.COCO_JSON_SKELETON


def test_create_batches_equals_the_reference(fx):
    from effocr_amd.pipeline import create_batches, iteration
    for c in fx["create_batches"]:
        data = [None if i in c["none_at"] else torch.full((3, 224, 224), float(i + 1)) for i in range(c["n"])]
        out = create_batches(data) if c["batch_size"] == 64 else create_batches(data, batch_size=c["batch_size"])
        assert [list(b.shape) for b in out] == c["shapes"], c
        assert [str(b.dtype) for b in out] == c["dtypes"]
        assert [[float(v) for v in b[:, 0, 0, 0]] for b in out] == c["row_first"]
        assert [bool((b == b[:, :1, :1, :1]).all()) for b in out] == c["row_is_constant"]
        assert all(isinstance(b, np.ndarray) for b in out)
    it = fx["iteration"]

    class M:
        def run(self, x):
            return [x * 2]
    out = iteration(M(), np.arange(6, dtype=np.float32))
    assert isinstance(out, tuple) == it["is_tuple"] and len(out) == it["len"] and (out[0] is out[1]) == it["same_object"]
    assert len(out[0]) == it["inner_len"] and out[0][0].tolist() == it["value"]


def test_onnx_driver_en_preprocess_equals_the_reference(fx):
    """infer_effocr_onnx_multi.py:70-90 — product: stable sort + ``word_end_indices``; oracle: run_effocr_ref.en_preprocess."""
    from effocr_amd.pipeline import word_end_indices
    from oracle import run_effocr_ref as R
    for c in fx["multi_en_preprocess"]:
        chars, words, vertical = np.asarray(c["chars"], np.float32).reshape(-1, 4), np.asarray(c["words"], np.float32).reshape(-1, 4), c["vertical"]
        axis = 1 if vertical else 0
        sc = chars[np.argsort(chars[:, axis], kind="stable")]
        sw = words[np.argsort(words[:, axis], kind="stable")]
        assert _floats(sc) == c["sorted_chars"]
        assert word_end_indices(sc[:, 2].tolist(), sw[:, 0].tolist()) == c["word_end_idx"], c
        s, wei = R.en_preprocess(torch.from_numpy(chars), torch.from_numpy(words), vertical=vertical)
        assert _floats(s) == c["sorted_chars"] and [int(i) for i in wei] == c["word_end_idx"]
    for c in fx["multi_jp_preprocess"]:
        chars = np.asarray(c["chars"], np.float32).reshape(-1, 4)
        assert _floats(chars[np.argsort(chars[:, 1 if c["vertical"] else 0], kind="stable")]) == c["sorted_chars"]


def _check_post(fn, c):
    if "raises" in c:
        with pytest.raises((AssertionError, IndexError)):
            fn()
    else:
        assert fn() == c["out"], c


def test_en_postprocess_equals_the_reference_both_drivers(fx):
    """infer_effocr_onnx_multi.py:92-131 (LARGE_NUMBER 1e9, anchor_multiplier argument) and infer_effocr.py:370-410."""
    from effocr_amd.postprocess import LinePostprocessor
    from oracle import postprocess_ref as R
    n_changed = 0
    for key in ("multi_en_postprocess", "single_en_postprocess"):
        for c in fx[key]:
            mult = c.get("anchor_multiplier", 4)
            post = LinePostprocessor(lang="en", anchor_margin=c["anchor_margin"], anchor_multiplier=mult)
            _check_post(lambda: post.en_postprocess(c["line"], c["word_end_idx"], c["heights"], c["bottoms"]), c)
            _check_post(lambda: R.en_postprocess(c["line"], c["word_end_idx"], c["heights"], c["bottoms"], anchor_margin=c["anchor_margin"],
                                                 anchor_multiplier=mult), c)
            n_changed += int(c.get("out") not in (None, c["line"]))
    assert n_changed > 200                                     # the fixtures really exercise spacing / case repair


def test_torch_driver_preprocess_equals_the_reference(fx):
    """EffOCR.en_preprocess / jp_preprocess, infer_effocr.py:345-367,412-418 (scores + thresholds, both result nestings)."""
    from effocr_amd.postprocess import LinePostprocessor
    from oracle import postprocess_ref as R
    for c in fx["single_en_preprocess"]:
        chars, words = np.asarray(c["chars"], np.float64).reshape(-1, 5), np.asarray(c["words"], np.float64).reshape(-1, 5)
        post = LinePostprocessor(lang="en", vertical=c["vertical"], score_thresh=c["score_thresh"], score_thresh_word=c["score_thresh_word"])
        s, wei = post.en_preprocess([[chars, words]] if c["wrapped"] else [chars, words])
        assert _floats(s) == c["sorted_chars"] and wei == c["word_end_idx"], c
        s2, wei2 = R.en_preprocess(chars, words, c["score_thresh"], c["score_thresh_word"], c["vertical"])
        assert _floats(s2) == c["sorted_chars"] and wei2 == c["word_end_idx"]
    for c in fx["single_jp_preprocess"]:
        post = LinePostprocessor(lang="jp", vertical=c["vertical"], score_thresh=c["score_thresh"])
        assert _floats(post.jp_preprocess([[np.asarray(c["chars"], np.float64).reshape(-1, 5)]])) == c["sorted_chars"]


def test_letterbox_geometry_and_box_helpers_equal_the_reference(fx, fa):
    """onnx_engines/localizer_engine.py:107-169: the geometry cv2.resize / cv2.copyMakeBorder were CALLED WITH by the reference."""
    from effocr_amd.localizer_engine import letterbox_geometry
    from oracle import yolo_ref as Y
    for c in fx["letterbox"]:
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c["kwargs"].items()}
        h, w = c["shape"]
        nh, nw, top, bottom, left, right, ratio, (dw, dh) = letterbox_geometry((h, w), **kw)
        assert [top, bottom, left, right] == c["border"], c
        assert [float(ratio[0]), float(ratio[1])] == c["ratio"] and float(dw) == c["dw"] and float(dh) == c["dh"]
        if c["resize"] is not None:
            assert [nw, nh] == c["resize"]
        else:
            assert (nw, nh) == (w, h)                           # the reference skipped cv2.resize: already that size
        if h * w <= 700 * 900 and max(h, w) <= 1000:            # the oracle's letterbox builds the pixels too: keep it small
            out, r2, (dw2, dh2) = Y.letterbox(np.zeros((h, w, 3), np.uint8), **kw)
            assert out.shape[:2] == (nh + top + bottom, nw + left + right) and [float(r2[0]), float(r2[1])] == c["ratio"]
            assert (float(dw2), float(dh2)) == (c["dw"], c["dh"])
        assert c["value"] == [114, 114, 114]
    x = torch.from_numpy(fa["xywh_in"])
    assert np.array_equal(Y.xywh2xyxy(x).numpy(), fa["xywh_out_torch"]) and np.array_equal(fa["xywh_out_torch"], fa["xywh_out_numpy"])
    # box_iou (:151-169): the oracle's NMS loop and the product's kernels compute IoU as inter / (area1 + area2 - inter + eps)
    a, b = torch.from_numpy(fa["iou_a"]), torch.from_numpy(fa["iou_b"])
    (a1, a2), (b1, b2) = a.unsqueeze(1).chunk(2, 2), b.unsqueeze(0).chunk(2, 2)
    inter = (torch.min(a2, b2) - torch.max(a1, b1)).clamp(0).prod(2)
    iou = inter / ((a2 - a1).prod(2) + (b2 - b1).prod(2) - inter + 1e-7)
    assert np.array_equal(iou.numpy(), fa["iou_out"])
    assert np.all(fa["iou_out"][:5, :5].diagonal() > 0.999999)


def test_medianpad_geometry_equals_the_reference(fx):
    """utils/datasets_utils.py:67-88: pad RIGHT and BOTTOM to a square; fill = the override (255,255,255 in create_paired_transform)
    or the per-channel median of the four borders."""
    from oracle.crop_transform_ref import pad_square_to_float
    seen_median = 0
    for c in fx["medianpad"]:
        h, w = c["shape"]
        l, t, r, b = c["padding_left_top_right_bottom"]
        assert (l, t) == (0, 0) and w + r == h + b == max(h, w)
        if c["override"] is not None:
            assert c["fill"] == c["override"] == [255, 255, 255]
            sq = pad_square_to_float(np.zeros((h, w, 3), np.uint8))
            assert sq.shape == (3, h + b, w + r)
            assert np.all(sq[:, h:, :] == 1.0) and np.all(sq[:, :, w:] == 1.0)
        else:
            im = np.asarray(c["image"], np.uint8)
            border = np.concatenate([im[:, w - 1, :], im[:, 0, :], im[0, :, :], im[h - 1, :, :]], axis=0)
            assert c["fill"] == [int(v) for v in np.median(border, axis=0)]
            seen_median += 1
    assert seen_median >= 2


# ---------------------------------------------------------------------------------------------------------------------
def line_image(seed, H, W):
    rng = np.random.RandomState(seed)
    cells = rng.randint(0, 8, ((H + 7) // 8, (W + 7) // 8, 3)).astype(np.uint8) * 32
    return np.ascontiguousarray(np.kron(cells, np.ones((8, 8, 1), np.uint8))[:H, :W])


def load_run_effocr_fixture(suffix=""):
    """suffix "": the miniature encoder (vit_tiny_test); "_vits": the same driver cases recorded over oracle A's ViT-S/16 (BASELINE
    configs[1]'s architecture; REF_ARCH=vit_small_patch16_224 python tests/golden/make_ref_golden.py)."""
    with open(os.path.join(G, f"ref_run_effocr{suffix}.json")) as f:
        meta = json.load(f)
    arrays = np.load(os.path.join(G, f"ref_run_effocr{suffix}.npz"))
    for c in meta["cases"]:
        c["images"], c["rows"] = [], []
        for l in c["lines"]:
            im = line_image(l["seed"], l["H"], l["W"])
            assert hashlib.sha256(im.tobytes()).hexdigest() == l["sha256"], "synthetic line generator drifted"
            c["images"].append(im)
            c["rows"].append(torch.from_numpy(arrays[l["rows"]]))
    return meta, arrays["index"]


def test_reference_run_effocr_replayed_through_the_oracle_driver():
    """The strings the REFERENCE's own run_effocr produced over oracle-backed engines (make_ref_golden.record_run_effocr) — the
    oracle's restated driver, fed the same NMS rows, must produce them character for character (en with and without case repair,
    jp, vertical; empty crops, negative / out-of-image coordinates, x.5 roundings, a line without boxes)."""
    from effocr_amd.weights import init_state_dict
    from oracle import run_effocr_ref as R
    meta, index = load_run_effocr_fixture()
    enc_sd = init_state_dict(meta["arch"], seed=meta["enc_seed"], img_size=meta["size"])
    total = 0
    for c in meta["cases"]:
        got, _ = R.run_effocr_ref(c["images"], None, meta["arch"], enc_sd, index, meta["chars"], c["lang"], vertical=c["vertical"],
                                  localizer_results=c["rows"], anchor_margin=c["anchor_margin"], size=meta["size"])
        assert got == c["outputs"], (c["lang"], c["vertical"], got, c["outputs"])
        assert c["coco"] == {"info": {"": ""}, "licenses": [{"": ""}], "images": [], "annotations": [], "categories": [{"id": 0, "name": "char"}]}
        total += sum(len(o) for o in c["outputs"] if o)
    assert total > 300


def infer_case_inputs(c):
    im = line_image(c["seed"], c["H"], c["W"])
    assert hashlib.sha256(im.tobytes()).hexdigest() == c["sha256"]
    cb, wb = np.asarray(c["chars"], np.float32).reshape(-1, 5), np.asarray(c["words"], np.float32).reshape(-1, 5)
    return im, ([cb, wb] if c["lang"] == "en" else [[cb]])


def test_reference_infer_replayed_through_the_oracle():
    """``EffOCR.infer`` (infer_effocr.py:255-343; kNN branch :310-319 with the default k = 10) as recorded from the reference over
    oracle-backed stages: the oracle's restated chain returns the same transcription, the same ten-neighbour strings per glyph and
    the same sorted / filtered boxes."""
    from effocr_amd.weights import init_state_dict
    from oracle import knn_ref
    from oracle.crop_transform_ref import paired_transform
    from oracle.encoders_ref import encoder_forward
    from oracle.postprocess_ref import infer_ref
    meta, index = load_run_effocr_fixture()
    enc_sd = init_state_dict(meta["arch"], seed=meta["enc_seed"], img_size=meta["size"])
    tf = lambda c: torch.from_numpy(np.asarray(paired_transform(c, size=meta["size"]), dtype=np.float32))
    enc = lambda x: encoder_forward(meta["arch"], enc_sd, x)
    knn = lambda q, k: torch.from_numpy(knn_ref.flat_ip_search(q.numpy(), index, k)[1])
    seen = 0
    for c in meta["infer"]:
        im, result = infer_case_inputs(c)
        out, nns, cb, wb = infer_ref(im, result, c["lang"], enc, knn, meta["chars"], tf, k=10, vertical=c["vertical"],
                                     anchor_margin=c["anchor_margin"])
        assert out == c["output"] and nns == c["output_nns"], (out, c["output"])
        assert (cb is None and c["char_bboxes"] is None) or _floats(cb) == c["char_bboxes"]
        seen += len(nns or [])
    assert seen > 35


# ---- the optional homoglyph spell checker (infer_effocr.py:401-402, utils/spell_check_utils.py), recorded by make_ref_spellcheck.py ----
@pytest.fixture(scope="module")
def sx():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_spellcheck.json"), encoding="utf-8") as f:
        return json.load(f)


def test_spell_check_tables_and_helpers_equal_the_reference(sx):
    from effocr_amd import spell_check as S
    assert S.create_homoglyph_dict() == sx["homoglyphs"]
    assert sorted(S.create_common_abbrev()) == sx["abbrevs"]
    words, abbr = sx["words"], set(sx["abbrevs"])
    for c in sx["helpers"]:
        s = c["s"]
        got = {"s": s, "depunctuate": S.depunctuate(s), "is_number": S.is_number(s), "is_word": S.is_word(s, words),
               "is_initial": S.is_initial(s), "all_caps": S.all_caps(s), "is_abbrev": S.is_abbrev(s, abbr)}
        assert got == c
    for c in sx["majority"]:
        if "raises" in c:
            with pytest.raises(IndexError):
                S.majority_normalize(c["s"], sx["homoglyphs"])
        else:
            assert S.majority_normalize(c["s"], sx["homoglyphs"]) == c["out"], c


def test_visual_spell_checker_equals_the_reference(sx):
    """192 recorded calls of the reference's visual_spell_checker over a synthetic dictionary: dictionary hits by frequency, abbreviations,
    initials, numbers, splitters (incl. the double-quote quirk), the H -> ll index shift, a beam of 3, majority normalisation on / off."""
    from effocr_amd import spell_check as S
    words, sim, abbr = sx["words"], sx["homoglyphs"], set(sx["abbrevs"])
    changed = 0
    for c in sx["checker"]:
        assert "raises" not in c
        got = S.visual_spell_checker(c["line"], words, sim, abbr, beam=c["beam"], majority_norm=c["majority_norm"])
        assert got == c["out"], c
        changed += got != c["line"]
    assert changed > 60


def test_en_postprocess_with_spell_check_equals_the_reference(sx):
    """EffOCR.en_postprocess(spell_check=True): the case-repair index lists are computed before the correction and applied after it."""
    from effocr_amd.postprocess import LinePostprocessor
    words, sim, abbr = sx["words"], sx["homoglyphs"], set(sx["abbrevs"])
    for c in sx["postprocess"]:
        post = LinePostprocessor(lang="en", anchor_margin=c["anchor_margin"], spell_check=True, worddict=words, simdict=sim, abbrevset=abbr)
        _check_post(lambda: post.en_postprocess(c["line"], c["word_end_idx"], c["heights"], c["bottoms"]), c)


def test_load_worddict_reads_the_symspellpy_format(tmp_path):
    from effocr_amd import spell_check as S
    p = tmp_path / "freq.txt"
    p.write_text("the 23135851162\nhello 32960381\nmr 5000\nbroken line here\n\nworld 1\n", encoding="utf-8")
    d = S.load_worddict(str(p))
    assert d == {"the": 23135851162, "hello": 32960381, "world": 1}          # "mr" removed: a de-punctuated common abbreviation
    assert S.create_worddict(str(p)) == d
