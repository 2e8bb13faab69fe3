"""GPU parity: effocr_crop_transform (device-side create_paired_transform) vs oracle/crop_transform_ref.py.

Tolerances (floating point, stated per mode; outputs are ImageNet-normalised, range about [-2.2, 2.7]):
  antialias=True  : 2e-5 absolute — same weights, taps and accumulation order as ATen, differences are
                    fma contraction and the position of the /255;
  antialias=False : 2e-6 * L absolute (L = padded side) — the tap weight is a difference of fp32 pixel
                    coordinates of magnitude L, whose rounding differs between evaluation orders (torch's own
                    CPU and GPU kernels disagree by the same amount).
"""
import numpy as np
import pytest
import torch

from oracle import crop_transform_ref as R

pytestmark = pytest.mark.gpu


def _image(h, w, seed):
    rng = np.random.default_rng(seed)
    # smooth background + glyph-like dark strokes + noise: exercises both flat and high-frequency regions
    yy, xx = np.mgrid[0:h, 0:w]
    img = 200 + 40 * np.sin(xx / 7.0)[..., None] * np.cos(yy / 5.0)[..., None] + rng.normal(0, 10, (h, w, 3))
    for _ in range(40):
        x, y = rng.integers(0, w - 4), rng.integers(0, h - 4)
        img[y:y + rng.integers(2, 30), x:x + rng.integers(1, 4)] = rng.integers(0, 60)
    return np.clip(img, 0, 255).astype(np.uint8)


BOXES = [(3, 5, 40, 61), (0, 0, 300, 180), (100, 20, 131, 52), (10, 10, 11, 11), (0, 50, 300, 53), (250, 0, 253, 180),
         (17, 3, 290, 170), (5, 5, 229, 229 - 50), (60, 60, 60 + 224, 60 + 112), (-40, 10, 300, 100), (20, -30, 90, 400)]


@pytest.mark.parametrize("antialias", [True, False])
@pytest.mark.parametrize("size", [224, 32])
def test_boxes_match_oracle(dev, antialias, size):
    from effocr_amd.transforms import PairedTransform, slice_boxes
    img = _image(180, 300, 1)
    t = PairedTransform(size=size, antialias=antialias, device=dev)
    got = t.boxes(img, BOXES, already_int=True).cpu().numpy()
    ref = R.transform_boxes(img, BOXES, size=size, antialias=antialias)
    assert got.shape == ref.shape == (len(BOXES), 3, size, size)
    ib = slice_boxes(BOXES, 180, 300)
    for i in range(len(BOXES)):
        L = max(ib[i, 2] - ib[i, 0], ib[i, 3] - ib[i, 1])
        tol = 2e-5 if antialias else max(2e-5, 2e-6 * L) * 4.5     # 1/std <= 4.5
        err = np.abs(got[i] - ref[i]).max()
        assert err <= tol, (i, BOXES[i], err, tol)


def test_numpy_restatement_and_torch_agree_with_device(dev):
    """Large down-scale (L = 900 -> 224, 9-tap antialias window) and up-scale from 2 px."""
    from effocr_amd.transforms import PairedTransform
    img = _image(900, 700, 2)
    boxes = [(0, 0, 700, 900), (10, 10, 12, 12), (100, 0, 140, 900)]
    t = PairedTransform(device=dev)
    got = t.boxes(img, boxes, already_int=True).cpu().numpy()
    ref = R.transform_boxes(img, boxes)
    assert np.abs(got - ref).max() <= 2e-5
    ref_np = R.transform_boxes(img, boxes[1:2], use_torch=False)
    assert np.abs(got[1:2] - ref_np).max() <= 2e-5


def test_float_boxes_round_half_even_and_per_crop_call(dev):
    from effocr_amd.transforms import create_paired_transform
    img = _image(64, 96, 3)
    t = create_paired_transform(device=dev)
    fb = [(2.5, 3.5, 40.49, 50.5, 0.99), (0.4, 0.6, 95.5, 63.5, 0.5)]      # -> (2,4,40,50), (0,1,96,64)
    got = t.boxes(img, fb).cpu().numpy()
    ref = R.transform_boxes(img, [R.round_box(b[:4]) for b in fb])
    assert np.abs(got - ref).max() <= 2e-5
    one = t(img[4:50, 2:40, :])                                            # reference convention: crop -> [3,224,224]
    assert tuple(one.shape) == (3, 224, 224) and one.is_cuda
    assert np.abs(one.cpu().numpy() - ref[0]).max() <= 2e-5


def test_errors(dev):
    from effocr_amd.transforms import PairedTransform
    img = _image(40, 40, 4)
    t = PairedTransform(device=dev)
    with pytest.raises(ValueError):
        t.boxes(img, [(10, 10, 10, 30)], already_int=True)                 # empty: the reference dies on it too
    with pytest.raises(ValueError):
        t.boxes(img[..., :2], [(0, 0, 4, 4)], already_int=True)
    with pytest.raises(ValueError):
        PairedTransform(size=30)
    assert t.boxes(img, []).shape == (0, 3, 224, 224)


def test_transform_feeds_encoder_end_to_end(dev):
    """image + boxes -> crops -> encoder -> ids equals the oracle chain on the same crops (resnet18, fp32)."""
    from effocr_amd.encoders import HipEncoder
    from effocr_amd.transforms import PairedTransform
    from effocr_amd.weights import init_state_dict
    from oracle.encoders_ref import encoder_forward
    img = _image(120, 400, 5)
    boxes = [(x, 10, x + 28, 70) for x in range(5, 360, 30)]
    x = PairedTransform(size=32, device=dev).boxes(img, boxes, already_int=True)
    sd = init_state_dict("resnet18", seed=0)
    emb = HipEncoder("resnet18", sd, img_size=32, precision="fp32", device=dev).forward(x, normalize=False).cpu()
    ref = encoder_forward("resnet18", sd, torch.from_numpy(R.transform_boxes(img, boxes, size=32)))
    assert ((emb - ref).abs().max() / ref.abs().max()).item() <= 1e-4


def test_recognize_boxes_end_to_end(dev):
    """Recognizer.recognize_boxes = infer_effocr.py:281-319,337-338 with the crop loop on the device:
    the index holds the embeddings of the same glyph boxes, so every box must retrieve its own character,
    and the neighbour lists must equal the oracle chain (CPU transform -> CPU encoder -> flat_ip_search)."""
    from effocr_amd.encoders import HipEncoder
    from effocr_amd.knn import FaissKNN, IndexFlatIP
    from effocr_amd.pipeline import Recognizer
    from effocr_amd.weights import init_state_dict
    from oracle import knn_ref
    from oracle.encoders_ref import encoder_forward, l2_normalize
    img = _image(80, 640, 6)
    boxes = [(x + 0.3, 8.6, x + 30.2, 70.4, 0.9) for x in range(4, 600, 31)]
    chars = [chr(0x3041 + i) for i in range(len(boxes))]
    sd = init_state_dict("resnet18", seed=1)
    enc = HipEncoder("resnet18", sd, img_size=32, precision="fp32", device=dev)
    # reference convention (infer_effocr.py:226,286-291): double_clipped is hard-coded True -> every crop spans the whole
    # line height: (x0, 0, x1, H) after rounding.  The index is built from exactly those crops on the CPU.
    H = img.shape[0]
    clipped = [(R.round_box(b[:4])[0], 0, R.round_box(b[:4])[2], H) for b in boxes]
    ref_crops = R.transform_boxes(img, clipped, size=32)
    ref_emb = l2_normalize(encoder_forward("resnet18", sd, torch.from_numpy(ref_crops))).numpy()
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False)
    knn.train(torch.from_numpy(ref_emb).to(dev))
    rec = Recognizer(enc, knn, chars, knn=5)
    nearest, nns, out = rec.recognize_boxes(img, boxes)
    assert out == "".join(chars) and [n[0] for n in nearest] == chars
    _, I_ref = knn_ref.flat_ip_search(ref_emb, ref_emb, 5)
    same = sum([chars[j] for j in I_ref[i]] == nearest[i] for i in range(len(boxes)))
    assert same >= len(boxes) - 1                      # fp32 GPU vs CPU embeddings: at most one near-tie may swap
    assert rec.recognize_boxes(img, []) == ([], [], "")
    # the flag really changes the crops: tight boxes give different embeddings (not the self-retrieval above) ...
    from effocr_amd.transforms import PairedTransform
    tf = PairedTransform(size=32, device=dev)
    tight = tf.boxes(img, boxes)
    full = tf.boxes(img, clipped, already_int=True)
    assert not torch.equal(tight, full)
    np.testing.assert_allclose(full.cpu().numpy(), ref_crops, atol=3e-5)
    # ... and vertical lines are clipped the other way: (0, y0, W, y1)
    vb = [(10.2, y + 0.4, 50.7, y + 24.6) for y in range(2, 50, 25)]
    W_ = img.shape[1]
    n_v, _, _ = rec.recognize_boxes(img, vb, vertical=True)
    want = tf.boxes(img, [(0, R.round_box(b)[1], W_, R.round_box(b)[3]) for b in vb], already_int=True)
    got_idx = rec.neighbors(want)[1][:, 0].cpu().tolist()
    assert [n[0] for n in n_v] == [chars[i] for i in got_idx]


def test_line_recognizer_en_end_to_end(dev):
    """infer_effocr.py:268-343 on the device path: localizer result (char + word boxes) -> crops -> encoder -> kNN ->
    en_postprocess.  Each box retrieves its own glyph (index = the boxes' own embeddings); the word boxes put spaces."""
    from effocr_amd.encoders import HipEncoder
    from effocr_amd.knn import FaissKNN, IndexFlatIP
    from effocr_amd.pipeline import Recognizer
    from effocr_amd.postprocess import LinePostprocessor, LineRecognizer
    from effocr_amd.transforms import PairedTransform
    from effocr_amd.weights import init_state_dict
    img = _image(60, 400, 9)
    text = "theCATran"
    xs = [5 + 40 * i for i in range(len(text))]
    chars_b = np.array([[x, 10, x + 30, 50, 0.9] for x in xs] + [[380, 10, 395, 50, 0.1]], dtype=np.float32)[::-1].copy()   # reversed + a low-score box
    words_b = np.array([[3, 8, 116, 52, 0.9], [124, 8, 236, 52, 0.9], [244, 8, 356, 52, 0.9]], dtype=np.float32)
    sd = init_state_dict("resnet18", seed=2)
    enc = HipEncoder("resnet18", sd, img_size=32, precision="fp32", device=dev)
    tf = PairedTransform(size=32, device=dev)
    emb = enc.forward(tf.boxes(img, [(x, 0, x + 30, 60) for x in xs], already_int=True), normalize=True)   # double-clipped crops (reference default)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False)
    knn.train(emb)
    rec = Recognizer(enc, knn, list(text), knn=3)
    lr = LineRecognizer(rec, LinePostprocessor(lang="en"), char_transform=tf)
    out, nns, cb, wb = lr.infer(img, (chars_b, words_b))
    assert out == "the CAT ran" and len(nns) == len(text) and len(cb) == len(text) and wb.shape == (3, 5)
    jp = LineRecognizer(rec, LinePostprocessor(lang="jp"), char_transform=tf)
    out_jp, _, cb_jp, wb_jp = jp.infer(img, [[chars_b]])
    assert out_jp == text and wb_jp is None
    assert jp.infer(img, [[chars_b[:1] * 0]]) == (None, None, None, None)            # nothing above the score threshold


def test_device_transform_matches_committed_golden(dev):
    """tests/golden/crop_transform.npz (make_golden.py: oracle output, itself cross-checked against torch interpolate):
    the HIP kernel against the COMMITTED vectors, both resize flavours, no oracle import on this path."""
    import os
    from effocr_amd.transforms import PairedTransform
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "crop_transform.npz"))
    img, boxes, size = g["image"], [tuple(int(v) for v in b) for b in g["boxes"]], int(g["size"])
    for aa, key in ((True, "out_aa"), (False, "out_plain")):
        got = PairedTransform(size=size, device=dev, antialias=aa).boxes(img, boxes, already_int=True).cpu().numpy()
        assert got.shape == g[key].shape
        np.testing.assert_allclose(got, g[key], atol=3e-5 if aa else 2e-4)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_16bit_hand_off_is_the_rounded_fp32_result(dev, dtype):
    """SURVEY f-2 ("uint8 line image + box list -> [B,3,224,224] bf16"): effocr_crop_transform_batch_ex with a 16-bit output type writes
    exactly the fp32 result rounded once (round to nearest even) — bit for bit torch's own cast of the fp32 kernel's output — for
    both resize flavours, including the zero crop of an empty box; and against the committed golden vectors within the rounding."""
    import os
    from effocr_amd.transforms import PairedTransform, slice_boxes
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "crop_transform.npz"))
    img, size = g["image"], int(g["size"])
    H, W = img.shape[:2]
    ib = slice_boxes([tuple(int(v) for v in b) for b in g["boxes"]], H, W)
    boxes5 = np.concatenate([ib, np.zeros((len(ib), 1), np.int32)], 1)
    boxes5 = np.concatenate([boxes5, np.array([[5, 5, 5, 9, 0], [0, 0, 10, 10, 3]], np.int32)])      # an empty box, a box naming no image
    stack = torch.from_numpy(np.ascontiguousarray(img))[None].to(dev)
    b5 = torch.from_numpy(boxes5).to(dev)
    for aa, key in ((True, "out_aa"), (False, "out_plain")):
        tf = PairedTransform(size=size, device=dev, antialias=aa)
        f32 = tf.boxes_batch(stack, b5)
        h16 = tf.boxes_batch(stack, b5, dtype=dtype)
        assert h16.dtype == dtype and h16.shape == f32.shape
        assert torch.equal(h16, f32.to(dtype))                                   # 0 ulp from the rounded fp32 result
        assert (h16[-2:] == 0).all()
        ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)                     # half an ulp at magnitude <= 4, plus the fp32 kernel's own bound
        np.testing.assert_allclose(h16[:len(ib)].float().cpu().numpy(), g[key], atol=(3e-5 if aa else 2e-4) + 2 * ulp)
    with pytest.raises(ValueError):
        tf.boxes_batch(stack, b5, dtype=torch.float64)
