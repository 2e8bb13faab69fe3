"""-m gpu: the product-level ``run_effocr`` (effocr_amd/pipeline.py; infer_effocr_onnx_multi.py:227-397) — BASELINE configs[4]
on one GPU at its own line size (4096 x 256) — against the CPU restatement of the whole driver (oracle/run_effocr_ref.py:
yolo_ref -> box parsing -> crop_transform_ref -> encoders_ref -> flat_ip.c -> postprocess_ref)."""
import numpy as np
import pytest
import torch

from effocr_amd.knn import FaissKNN, IndexFlatIP
from effocr_amd.localizer_engine import EffLocalizer, init_yolov5s_state_dict
from effocr_amd.pipeline import run_effocr, word_end_indices
from effocr_amd.recognizer_engine import EffRecognizer
from effocr_amd.transforms import PairedTransform
from effocr_amd.weights import init_state_dict
from oracle import run_effocr_ref as R

pytestmark = pytest.mark.gpu

CHARS = list("aenrwuosvcxzTHEQUICKBROWN-") + [chr(0x4E00 + i) for i in range(200)]


def _busy(nc=2, seed=0, obj=5.5, cls=(2.5, 2.4)):        # ~50 character boxes and a few word boxes per 4096 x 256 line at conf 0.5
    sd = init_yolov5s_state_dict(nc, seed=seed)
    for l in range(3):
        b = sd[f"model.24.m.{l}.bias"].view(3, nc + 5)
        b[:, 4] += obj
        b[:, 5] += cls[0]
        b[:, 6] += cls[1]
    return sd


def _lines(n, H, W, seed):
    rng = np.random.default_rng(seed)
    return [(rng.integers(0, 256, (H, W, 3)) // 32 * 32).astype(np.uint8) for _ in range(n)]


def _setup(dev, arch="vit_small_patch16_224", size=224, precision="fp32", conf=0.5, iou=0.05, seed=0):
    loc_sd = _busy(2, seed=seed)
    loc = EffLocalizer(loc_sd, iou_thresh=iou, conf_thresh=conf, device=dev)
    enc_sd = init_state_dict(arch, seed=1, img_size=size)
    rec = EffRecognizer(enc_sd, arch=arch, precision=precision, img_size=size, device=dev)
    tf = PairedTransform(size=size, device=dev)
    return loc_sd, loc, enc_sd, rec, tf


def _match_boxes(got, ref, tol=0.05):
    """Device localizer vs the CPU oracle: the networks differ by fp32 summation order, so compare as sets with a tolerance."""
    assert abs(got.shape[0] - ref.shape[0]) <= max(2, ref.shape[0] // 50), (got.shape, ref.shape)
    matched = 0
    for r in ref:
        d = (got[:, :4] - r[:4]).abs().max(1)[0] + (got[:, 5] != r[5]).float() * 1e3
        matched += int(d.min().item() < tol)
    assert matched >= ref.shape[0] - max(2, ref.shape[0] // 50)


@pytest.mark.parametrize("lang", ["en", "jp"])
def test_run_effocr_c5_lines_against_the_oracle_chain(dev, lang):
    """Three 4096 x 256 line images (BASELINE configs[4]) through the product function and through the CPU driver restatement.
    Stage 1 (localizer): box sets agree within 0.05 px (set comparison: a box on a threshold may differ).  Stages 2-5 (box parsing,
    scaling / double clipping / numpy-slice semantics, crop transform, encoder (fp32 mode), k-NN, line assembly, en_postprocess):
    fed with the device localizer's own boxes, the oracle must produce the SAME strings, character for character.  The index is
    built from the oracle's embeddings of these very crops plus distractors, so every top-1 has a margin."""
    loc_sd, loc, enc_sd, rec, tf = _setup(dev, seed=2)
    lines = _lines(3, 256, 4096, seed=11)
    dev_results = loc.run(lines)                                            # CPU tensors [n,6] per line
    ora_results = R.localize(lines[:1], loc_sd, 0.5, 0.05)
    _match_boxes(dev_results[0], ora_results[0])
    n_char = [int((r[:, 5] == 0).sum()) for r in dev_results]
    assert all(5 <= n <= 400 for n in n_char), n_char
    # the oracle's embeddings of the crops the reference would cut -> glyph index (self-retrieval + distractors)
    crops = []
    for im, res in zip(lines, dev_results):
        cb = sorted(res[res[:, 5] == 0][:, :4], key=lambda x: x[0])
        for bb in cb:
            x0, _, x1, _ = torch.round(bb)
            x0, x1 = int(round(x0.item() * 4096 / 640)), int(round(x1.item() * 4096 / 640))
            c = im[0:256, x0:x1, :]
            crops.append(c if c.shape[1] > 0 else None)
    keep = [c for c in crops if c is not None][:48]
    from oracle.crop_transform_ref import paired_transform
    from oracle.encoders_ref import encoder_forward, l2_normalize
    xs = torch.stack([torch.from_numpy(np.asarray(paired_transform(c, size=224), dtype=np.float32)) for c in keep])
    emb = l2_normalize(encoder_forward("vit_small_patch16_224", enc_sd, xs)).numpy()
    rng = np.random.default_rng(5)
    distract = rng.standard_normal((len(CHARS) - emb.shape[0], emb.shape[1])).astype(np.float32)
    distract /= np.linalg.norm(distract, axis=1, keepdims=True)
    index = np.concatenate([emb, distract]).astype(np.float32)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False, device=dev)
    knn.train(torch.from_numpy(index))
    got, coco = run_effocr(lines, loc, rec, tf, lang, knn_func=knn, candidate_chars=CHARS, anchor_margin=0.15 if lang == "en" else None)
    assert coco == {"info": {"": ""}, "licenses": [{"": ""}], "images": [], "annotations": [], "categories": [{"id": 0, "name": "char"}]}
    want, _ = R.run_effocr_ref(lines, loc_sd, "vit_small_patch16_224", enc_sd, index, CHARS, lang, localizer_results=dev_results,
                               anchor_margin=0.15 if lang == "en" else None)
    assert list(got.keys()) == [0, 1, 2]
    for i in range(3):
        assert got[i] == want[i], (i, got[i], want[i])
    assert sum(len(w) for w in want if w) >= 15                             # the lines really carry text
    # second call, the BASELINE dtype (bf16 encoder): same boxes, same line structure; the characters agree wherever the top-1 has a
    # margin (the 48 indexed crops find themselves) — crops that are NOT in this synthetic index sit between random rows and may differ
    rec16 = EffRecognizer(enc_sd, arch="vit_small_patch16_224", precision="bf16", device=dev)
    got16, _ = run_effocr(lines, loc, rec16, tf, lang, knn_func=knn, candidate_chars=CHARS, anchor_margin=0.15 if lang == "en" else None)
    for i in range(3):
        assert len(got16[i]) == len(got[i])
        same = sum(a == b for a, b in zip(got16[i].lower(), got[i].lower()))
        assert same >= 0.85 * len(got[i]), (i, got16[i], got[i])


def test_run_effocr_strings_fp32_vs_bf16_operand_localizer(dev):
    """EffLocalizer(precision="bf16") rounds the operands of every activated convolution to bf16 (boxes within 0.3 px of the fp32
    network, tests/test_gpu_localizer.py).  STRING-level evidence on 16 lines of BASELINE configs[4]'s shape: the same lines through
    run_effocr with both localizers against one glyph index (the fp32 run's own crops + distractors, so every top-1 has a margin).
    Measured (round 4): on these lines — a random-weight detector firing densely on noise, i.e. many boxes sitting on the confidence /
    NMS thresholds — NONE of the 16 strings is identical: boxes appear / disappear and crop edges move by a pixel column.  The bf16-operand
    localizer therefore stays an opt-in and fp32 operands (the reference's arithmetic) the default; this test pins the default and
    records the agreement it sees (asserted only loosely: both runs read text of similar length)."""
    loc_sd, loc, enc_sd, rec, tf = _setup(dev, seed=2)
    loc16 = EffLocalizer(loc_sd, iou_thresh=0.05, conf_thresh=0.5, device=dev, precision="bf16")
    lines = _lines(16, 256, 4096, seed=21)
    res = loc.run(lines)
    crops = []
    for im, r in zip(lines, res):
        for bb in sorted(r[r[:, 5] == 0][:, :4], key=lambda x: x[0]):
            x0, _, x1, _ = torch.round(bb)
            x0, x1 = int(round(x0.item() * 4096 / 640)), int(round(x1.item() * 4096 / 640))
            if x1 > x0:
                crops.append(im[0:256, max(x0, 0):x1, :])
    crops = crops[:len(CHARS) - 30]
    from oracle.crop_transform_ref import paired_transform
    xs = np.stack([np.asarray(paired_transform(c, size=224), dtype=np.float32) for c in crops])
    emb = np.concatenate([rec.run(xs[i:i + 64])[0] for i in range(0, len(xs), 64)])
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    rng = np.random.default_rng(5)
    distract = rng.standard_normal((len(CHARS) - emb.shape[0], emb.shape[1])).astype(np.float32)
    distract /= np.linalg.norm(distract, axis=1, keepdims=True)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False, device=dev)
    knn.train(torch.from_numpy(np.concatenate([emb, distract]).astype(np.float32)))
    a, _ = run_effocr(lines, loc, rec, tf, "jp", knn_func=knn, candidate_chars=CHARS)
    b, _ = run_effocr(lines, loc16, rec, tf, "jp", knn_func=knn, candidate_chars=CHARS)
    same = [a[i] == b[i] for i in range(16)]
    pos = sum(x == y for i in range(16) for x, y in zip(a[i], b[i])) / max(1, sum(min(len(a[i]), len(b[i])) for i in range(16)))
    print(f"fp32- vs bf16-operand localizer: {sum(same)} / 16 strings identical, {100 * pos:.0f} % of the characters equal position by position")
    assert loc._eng_net.precision == "fp32" and EffLocalizer(loc_sd, device=dev)._eng_net.precision == "fp32"   # the default
    la, lb = sum(len(a[i]) for i in range(16)), sum(len(b[i]) for i in range(16))
    assert la >= 16 * 20 and abs(la - lb) <= 0.3 * la


def test_run_effocr_edge_cases(dev, tmp_path):
    """Mixed geometries (grouped), a path entry, a line without any box (en -> None as en_postprocess returns, jp -> ""),
    vertical text, the empty list, argument errors."""
    from PIL import Image
    loc_sd, loc, enc_sd, rec, tf = _setup(dev, arch="resnet18", size=32, precision="fp32", seed=3)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False, device=dev)
    g = torch.Generator().manual_seed(0)
    index = torch.nn.functional.normalize(torch.randn(len(CHARS), 512, generator=g), dim=1)
    knn.train(index)
    lines = _lines(2, 64, 640, seed=1) + _lines(1, 48, 400, seed=2)
    png = tmp_path / "l.png"
    Image.fromarray(lines[2]).save(png)
    mixed = [lines[0], str(png), lines[1]]
    got, _ = run_effocr(mixed, loc, rec, tf, "jp", knn_func=knn, candidate_chars=CHARS)
    want, _ = R.run_effocr_ref([lines[0], lines[2], lines[1]], loc_sd, "resnet18", enc_sd, index.numpy(), CHARS, "jp",
                               localizer_results=loc.run([lines[0], lines[2], lines[1]]), size=32)
    assert list(got.keys()) == [0, str(png), 2]
    assert [got[0], got[str(png)], got[2]] == want
    # vertical: sorted by y0, crops span the full width
    gotv, _ = run_effocr(lines[:2], loc, rec, tf, "jp", vertical=True, knn_func=knn, candidate_chars=CHARS)
    wantv, _ = R.run_effocr_ref(lines[:2], loc_sd, "resnet18", enc_sd, index.numpy(), CHARS, "jp", vertical=True,
                                localizer_results=loc.run(lines[:2]), size=32)
    assert [gotv[0], gotv[1]] == wantv
    # no detections at all
    quiet = EffLocalizer(init_yolov5s_state_dict(2, seed=0), iou_thresh=0.05, conf_thresh=0.99, device=dev)
    g0, _ = run_effocr(lines[:1], quiet, rec, tf, "en", knn_func=knn, candidate_chars=CHARS)
    g1, _ = run_effocr(lines[:1], quiet, rec, tf, "jp", knn_func=knn, candidate_chars=CHARS)
    assert g0 == {0: None} and g1 == {0: ""}
    assert run_effocr([], loc, rec, tf, "en", knn_func=knn, candidate_chars=CHARS)[0] == {}
    with pytest.raises(ValueError):
        run_effocr(lines[:1], loc, rec, tf, "fr", knn_func=knn, candidate_chars=CHARS)
    # localizer_output: the debug drawings of infer_effocr_onnx_multi.py:292-305 — one image per line, character regions outlined in red
    from PIL import Image
    gd, _ = run_effocr(lines[:2], loc, rec, tf, "en", localizer_output=str(tmp_path), knn_func=knn, candidate_chars=CHARS)
    assert gd == run_effocr(lines[:2], loc, rec, tf, "en", knn_func=knn, candidate_chars=CHARS)[0]      # drawing changes no result
    for j in (0, 1):
        drawn = np.array(Image.open(tmp_path / f"{j}.png").convert("RGB"))
        assert drawn.shape == lines[j].shape
        changed = (drawn != lines[j]).any(axis=2)
        assert changed.any() and (drawn[changed] == np.array([255, 0, 0])).all()            # only red outline pixels differ


def test_word_end_indices_keeps_the_reference_quirks():
    # word lefts 5 and 100: first word -> char whose right edge (10) is the closest one right of 5; second word has no candidate
    # right of 100 -> repeats the previous index (closest_idx is not reset, infer_effocr_onnx_multi.py:75-87)
    assert word_end_indices([10.0, 20.0, 30.0], [5.0, 100.0]) == [0, 0]
    assert word_end_indices([10.0, 20.0, 30.0], [15.0, 25.0]) == [1, 2]
    assert R.en_preprocess([torch.tensor([0., 0, 10, 5]), torch.tensor([12., 0, 20, 5]), torch.tensor([22., 0, 30, 5])],
                           [torch.tensor([15., 0, 40, 5]), torch.tensor([25., 0, 40, 5])])[1] == [1, 2]


class _PresetLocalizer:
    """Stands in for EffLocalizer.run_device with the NMS rows of a fixture (keyed by the image bytes)."""
    _model_backend = "yolo"

    def __init__(self, images, rows, dev):
        import hashlib
        self._h = hashlib
        self.dev = dev
        self.by_img = {hashlib.sha256(im.tobytes()).hexdigest(): r for im, r in zip(images, rows)}

    def run_device(self, imgs, max_det=1000):
        out = torch.zeros(len(imgs), max_det, 6, device=self.dev)
        cnt = torch.zeros(len(imgs), dtype=torch.int64, device=self.dev)
        for j, im in enumerate(imgs):
            r = self.by_img[self._h.sha256(im.cpu().numpy().tobytes()).hexdigest()]
            out[j, :r.shape[0]] = r.to(self.dev)
            cnt[j] = r.shape[0]
        return out, cnt


@pytest.mark.parametrize("suffix,precision", [("", "fp32"), ("_vits", "fp32"), ("_vits", "fp16"), ("_vits", "bf16")])
def test_run_effocr_reproduces_the_references_own_strings(dev, suffix, precision):
    """tests/golden/ref_run_effocr*.*: strings produced by the REFERENCE's run_effocr (infer_effocr_onnx_multi.py:227-397, imported and
    run by tests/golden/make_ref_golden.py over oracle-backed engines).  The product function — HIP crop transform, HIP encoder
    (fp32 mode), HIP k-NN, device box arithmetic, product line assembly — fed the same NMS rows must return them character for
    character: en with / without case repair, jp, vertical, empty crops (zero image), negative and out-of-image coordinates.
    "_vits" (round 5): the same cases recorded over the real-size ViT-S/16 (BASELINE configs[1]'s architecture), replayed in the exact
    fp32 mode AND in the engines' default fp16 mode (16-bit crop hand-off, fused kernels): the transcriptions must still be the
    reference's, character for character."""
    from test_ref_golden import load_run_effocr_fixture
    meta, index = load_run_effocr_fixture(suffix)
    enc_sd = init_state_dict(meta["arch"], seed=meta["enc_seed"], img_size=meta["size"])
    rec = EffRecognizer(enc_sd, arch=meta["arch"], precision=precision, img_size=meta["size"], device=dev)
    tf = PairedTransform(size=meta["size"], device=dev)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False, device=dev)
    knn.train(torch.from_numpy(index))
    total = 0
    for c in meta["cases"]:
        loc = _PresetLocalizer(c["images"], c["rows"], dev)
        got, coco = run_effocr(c["images"], loc, rec, tf, c["lang"], vertical=c["vertical"], knn_func=knn, candidate_chars=meta["chars"],
                               anchor_margin=c["anchor_margin"])
        assert [got[i] for i in range(len(c["images"]))] == c["outputs"], (c["lang"], c["vertical"], got, c["outputs"])
        assert coco == c["coco"]
        total += sum(len(o) for o in c["outputs"] if o)
    assert total > 300


@pytest.mark.parametrize("suffix", ["", "_vits"])
def test_line_recognizer_reproduces_the_references_infer(dev, suffix):
    """tests/golden/ref_run_effocr.json["infer"]: what the REFERENCE's ``EffOCR.infer`` (infer_effocr.py:255-343, kNN branch with the
    default k = 10) returned over oracle-backed stages.  The product chain — LinePostprocessor, HIP crop transform, HIP encoder (fp32),
    HIP IndexFlatIP top-10, ``indices_to_chars``, en_postprocess — returns the same transcription and boxes; the ten-neighbour strings
    are compared in full wherever the recorded score gaps between successive ranks exceed 2e-5 (fp32 summation order cannot reorder
    those), by their first character (the top-1, margin ~5e-3) otherwise."""
    from effocr_amd.encoders import AutoEncoderFactory
    from effocr_amd.pipeline import Recognizer
    from effocr_amd.postprocess import LinePostprocessor, LineRecognizer
    from test_ref_golden import infer_case_inputs, load_run_effocr_fixture
    meta, index = load_run_effocr_fixture(suffix)
    enc_sd = init_state_dict(meta["arch"], seed=meta["enc_seed"], img_size=meta["size"])
    enc = AutoEncoderFactory("timm", meta["arch"], precision="fp32", img_size=meta["size"])()
    enc.load_state_dict(enc_sd)
    enc.to(dev).eval()
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False)
    knn.train(torch.from_numpy(index))
    rec = Recognizer(enc, knn, meta["chars"], knn=10)
    full = 0
    for c in meta["infer"]:
        im, result = infer_case_inputs(c)
        post = LinePostprocessor(lang=c["lang"], vertical=c["vertical"], anchor_margin=c["anchor_margin"])
        out, nns, cb, wb = LineRecognizer(rec, post).infer(im, result)
        assert out == c["output"], (out, c["output"])
        if c["output_nns"] is None:
            assert nns is None and cb is None
            continue
        assert [[float(v) for v in b] for b in cb] == c["char_bboxes"]
        assert len(nns) == len(c["output_nns"])
        for got, want, gap in zip(nns, c["output_nns"], c["rank_gap"]):
            if gap > 2e-5:
                assert got == want
                full += 1
            else:
                assert got[:1] == want[:1]
    assert full >= 20


@pytest.mark.parametrize("vertical,axis", [(False, 0), (True, 1), (True, 0)])
def test_box_stage_kernel_equals_the_torch_expression(dev, vertical, axis):
    """csrc/boxes.hip (two launches) against the ~25 ATen launches it replaces (infer_effocr_onnx_multi.py:252-256,275-288,313-320):
    identical sorted boxes, character counts and crop slices — ties in the sort key (stable: NMS order), boxes whose scaled bounds are
    negative / beyond the line, half-way values for both roundings, lines with no characters, no rows at all, and every row a character."""
    from effocr_amd.pipeline import _char_boxes, _char_boxes_torch
    g = torch.Generator().manual_seed(5)
    L, max_det, H, W = 7, 1000, 256, 4096
    rows = torch.zeros(L, max_det, 6)
    rows[..., 0] = (torch.rand(L, max_det, generator=g) * 700 - 30)
    rows[..., 1] = (torch.rand(L, max_det, generator=g) * 700 - 30)
    rows[..., 2] = rows[..., 0] + torch.rand(L, max_det, generator=g) * 80
    rows[..., 3] = rows[..., 1] + torch.rand(L, max_det, generator=g) * 80
    rows[..., 4] = torch.rand(L, max_det, generator=g)
    rows[..., 5] = torch.randint(0, 2, (L, max_det), generator=g).float()
    rows[0, :300, 0] = torch.randint(0, 20, (300,), generator=g).float()            # many equal sort keys: stability
    rows[0, :300, 1] = torch.randint(0, 20, (300,), generator=g).float()
    rows[1, :200, :4] = torch.round(rows[1, :200, :4]) + 0.5                         # x.5: round half to even
    rows[2, :100, 0] = torch.arange(100).float() * 6.4 + 3.2                         # scaled values landing on .5 at W = 4096 (x 6.4)
    rows[3, :, 5] = 1                                                                # a line without characters
    rows[5, :, 5] = 0                                                                # every row a character
    counts = torch.tensor([300, 1000, 640, 50, 0, 1000, 17], dtype=torch.int32)
    rows, counts = rows.to(dev), counts.to(dev)
    b0, n0, x0 = _char_boxes_torch(rows, counts, max_det, H, W, axis, vertical)
    b1, n1, x1 = _char_boxes(rows, counts, max_det, H, W, axis, vertical)
    assert torch.equal(n0.cpu(), n1.cpu()) and n1.tolist()[3:5] == [0, 0] and n1.tolist()[5] == 1000
    assert torch.equal(x0.cpu(), x1.cpu()) and x1.dtype == torch.int32
    assert torch.equal(b0.cpu(), b1.cpu())
    e = _char_boxes(rows[:0], counts[:0], max_det, H, W, axis, vertical)
    assert e[2].shape == (0, 5)
