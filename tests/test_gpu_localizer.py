"""-m gpu: the YOLOv5 localizer engine (SURVEY 8 f-3 / BASELINE config 5) against oracle/yolo_ref.py — network, letterbox,
NMS and the EffLocalizer call convention of onnx_engines/localizer_engine.py:14-66."""
import numpy as np
import pytest
import torch

from effocr_amd.localizer_engine import EffLocalizer, HipLocalizer, init_yolov5s_state_dict, letterbox_geometry
from oracle import yolo_ref as Y

pytestmark = pytest.mark.gpu


def _busy_state_dict(nc=2, seed=0, obj_shift=5.5):
    """Seeded weights whose Detect head fires often enough to give the NMS something to do."""
    sd = init_yolov5s_state_dict(nc, seed=seed)
    for l in range(3):
        b = sd[f"model.24.m.{l}.bias"].view(3, nc + 5)
        b[:, 4] += obj_shift
        b[:, 5:] += 1.5
    return sd


@pytest.mark.parametrize("shape,B", [((640, 640), 1), ((64, 96), 3), ((320, 256), 2)])
def test_network_matches_oracle(dev, shape, B):
    """Every layer type of the graph (stem im2col, strided / 1x1 / 3x3 convs + folded BN + SiLU, C3 with and without
    shortcuts writing concat slices, SPPF pool chain, upsample into slices, Detect decode) in one comparison."""
    sd = _busy_state_dict(2, seed=1)
    eng = HipLocalizer(sd, input_shape=shape, device=dev)
    x = torch.rand(B, 3, *shape, generator=torch.Generator().manual_seed(3))
    got = eng.forward(x.to(dev)).cpu()
    ref = Y.yolov5s_forward(sd, x)
    assert got.shape == ref.shape == (B, eng.num_predictions, 7)
    assert eng.num_predictions == 3 * sum((shape[0] // s) * (shape[1] // s) for s in (8, 16, 32))
    err = (got - ref).abs()
    # boxes in pixels (scale: the input size), probabilities in [0, 1]; fp32 MFMA vs the CPU's conv summation order
    assert (err[..., :4].max() / ref[..., :4].abs().max()).item() < 2e-4
    assert err[..., 4:].max().item() < 2e-4
    assert torch.equal(got, eng.forward(x.to(dev)).cpu())                 # deterministic
    eng.set_option("direct_stem", 0)                                      # the stem through im2col + implicit GEMM instead of the direct kernel
    alt = eng.forward(x.to(dev)).cpu()
    err = (alt - ref).abs()
    assert (err[..., :4].max() / ref[..., :4].abs().max()).item() < 2e-4 and err[..., 4:].max().item() < 2e-4
    assert (alt - got).abs().max().item() < 1e-2 * max(shape)             # two summation orders of the same layer


def test_stem_kernels_agree_bit_for_bit(dev):
    """The stem has two direct kernels: scalar-loaded weights + 8-byte input pairs (4 pixels per thread; taken when the input pointer is
    8-byte aligned) and the LDS-weight kernel (2 pixels per thread; the fallback).  Same taps in the same fmaf order per output, so an input
    that starts 4 bytes off an 8-byte boundary — a view into a larger buffer — must give the same bits as the aligned copy."""
    sd = _busy_state_dict(2, seed=4)
    shape, B = (96, 160), 2
    eng = HipLocalizer(sd, input_shape=shape, device=dev)
    x = torch.rand(B, 3, *shape, generator=torch.Generator().manual_seed(5)).to(dev)
    buf = torch.empty(x.numel() + 1, dtype=torch.float32, device=dev)
    off = 1 if buf.data_ptr() % 8 == 0 else 0                            # first element of the view at 4 (mod 8)
    view = buf[off:off + x.numel()].view_as(x)
    view.copy_(x)
    assert view.data_ptr() % 8 == 4 and x.data_ptr() % 8 == 0 and view.is_contiguous()
    a = eng.forward(x).cpu()
    b = eng.forward(view).cpu()
    assert torch.equal(a, b)


@pytest.mark.parametrize("shape,B", [((640, 640), 1), ((64, 96), 3)])
def test_network_bf16_operands(dev, shape, B):
    """precision="bf16": the convolutions with an activation run on bf16-rounded operands (fp32 accumulation, fp32 Detect heads).
    Against oracle A (fp32): boxes within a pixel, probabilities within 5e-3; deterministic; and it really is another path."""
    sd = _busy_state_dict(2, seed=1)
    eng = HipLocalizer(sd, input_shape=shape, device=dev, precision="bf16")
    x = torch.rand(B, 3, *shape, generator=torch.Generator().manual_seed(3))
    got = eng.forward(x.to(dev)).cpu()
    ref = Y.yolov5s_forward(sd, x)
    err = (got - ref).abs()
    assert err[..., :4].max().item() < 1.0 and err[..., :4].mean().item() < 0.05
    assert err[..., 4:].max().item() < 5e-3
    assert torch.equal(got, eng.forward(x.to(dev)).cpu())
    exact = HipLocalizer(sd, input_shape=shape, device=dev).forward(x.to(dev)).cpu()
    assert not torch.equal(got, exact)
    eng.set_option("bf16_operands", 0)                                    # the switch goes both ways on one handle
    assert torch.equal(eng.forward(x.to(dev)).cpu(), exact)
    with pytest.raises(ValueError):
        HipLocalizer(sd, input_shape=shape, device=dev, precision="fp16")


def test_single_class_model_and_validation(dev):
    sd = init_yolov5s_state_dict(1, seed=2)                               # 3 * 6 = 18 head channels -> padded to 20 inside
    eng = HipLocalizer(sd, input_shape=(64, 64), device=dev)
    x = torch.rand(1, 3, 64, 64)
    got, ref = eng.forward(x.to(dev)).cpu(), Y.yolov5s_forward(sd, x)
    assert got.shape == (1, 252, 6) and (got - ref).abs().max().item() < 2e-3
    with pytest.raises(ValueError):
        eng.forward(torch.rand(1, 3, 32, 64, device=dev))
    bad = dict(sd)
    bad.pop("model.9.cv2.bn.running_var")
    with pytest.raises(ValueError):
        HipLocalizer(bad, input_shape=(64, 64), device=dev)
    with pytest.raises(Exception):
        HipLocalizer(sd, input_shape=(70, 64), device=dev)                # not a multiple of the stride


@pytest.mark.parametrize("hw", [(50, 300), (256, 4096), (480, 640), (640, 640), (1000, 37), (31, 33)])
@pytest.mark.parametrize("bgr", [False, True])
def test_letterbox_matches_oracle_exactly(dev, hw, bgr):
    """Fixed-point bilinear + 114 border + channel order + /255: bit-identical to the restated cv2 arithmetic."""
    sd = init_yolov5s_state_dict(2, seed=0)
    eng = HipLocalizer(sd, input_shape=(640, 640), device=dev)
    im = np.random.default_rng(hw[0] + hw[1]).integers(0, 256, hw + (3,)).astype(np.uint8)
    got = eng.letterbox(im, bgr=bgr).cpu().numpy()
    ref = Y.load_localizer_img(im, (640, 640), bgr=bgr)
    assert got.shape == ref.shape == (1, 3, 640, 640)
    assert np.array_equal(got, ref)


def _random_pred(n, nc, seed, dup=True):
    g = torch.Generator().manual_seed(seed)
    p = torch.zeros(n, 5 + nc)
    p[:, 0:2] = torch.rand(n, 2, generator=g) * 600 + 20
    p[:, 2:4] = torch.rand(n, 2, generator=g) * 60 + 4
    p[:, 4] = torch.rand(n, generator=g)
    p[:, 5:] = torch.rand(n, nc, generator=g)
    if dup and n >= 40:
        p[10:20] = p[0:10]                                 # exact duplicates: equal confidences, IoU = 1
        p[30:35, 4] = p[25:30, 4]
        p[30:35, 5:] = p[25:30, 5:]                        # equal confidences on different boxes
    return p


@pytest.mark.parametrize("n,nc,conf,iou,max_det", [(2000, 2, 0.3, 0.01, 1000), (2000, 2, 0.05, 0.45, 1000), (5000, 3, 0.01, 0.6, 300),
                                                   (300, 1, 0.5, 0.2, 5), (64, 2, 0.999, 0.5, 10), (1, 2, 0.0, 0.5, 10), (25200, 2, 0.6, 0.3, 1000)])   # (full-size case: conf 0.6 keeps the ORACLE's Python loop at seconds — 0.2 took 110 s of the suite)
def test_nms_matches_oracle_exactly(dev, n, nc, conf, iou, max_det):
    """Same prediction tensor into the device NMS and the restated non_max_suppression: identical rows, order and count
    (class offsets, duplicate boxes, tied confidences, the max_det cut, the empty result)."""
    sd = init_yolov5s_state_dict(nc, seed=0)
    eng = HipLocalizer(sd, input_shape=(64, 64), device=dev)
    pred = _random_pred(n, nc, seed=n + nc)
    got = eng.nms(pred.to(dev), conf, iou, max_det=max_det).cpu()
    ref = Y.non_max_suppression(pred[None], conf, iou, max_det=max_det)[0]
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.equal(got, ref)
    agn = eng.nms(pred.to(dev), conf, iou, max_det=max_det, agnostic=True).cpu()
    assert torch.equal(agn, Y.non_max_suppression(pred[None], conf, iou, agnostic=True, max_det=max_det)[0])


@pytest.mark.parametrize("B,n,nc,conf,iou,max_det", [(3, 2000, 2, 0.05, 0.45, 64), (3, 25200, 2, 0.01, 0.05, 64), (2, 300, 1, 0.5, 0.2, 5),
                                                     (2, 64, 2, 0.999, 0.5, 10), (1, 1, 2, 0.0, 0.5, 10), (1, 25600, 3, 0.75, 0.6, 128),
                                                     (2, 2000, 2, 0.05, 0.45, 1000), (2, 25601, 2, 0.97, 0.3, 16)])
def test_nms_batch_matches_oracle_exactly(dev, B, n, nc, conf, iou, max_det):
    """effocr_nms_batch — the one-launch greedy kernel (max_det <= 128, n <= 25600) and its fall-back to the per-image kernels —
    against the restated non_max_suppression, image by image: identical rows, order and counts (duplicates, tied confidences,
    a detector that fires on every anchor, the max_det cut)."""
    sd = init_yolov5s_state_dict(nc, seed=0)
    eng = HipLocalizer(sd, input_shape=(64, 64), device=dev)
    pred = torch.stack([_random_pred(n, nc, seed=7 * b + n + nc) for b in range(B)])
    for agnostic in (False, True):
        rows, cnt = eng.nms_batch_async(pred.to(dev), conf, iou, max_det=max_det, agnostic=agnostic)
        rows, cnt = rows.cpu(), cnt.cpu().tolist()
        ref = Y.non_max_suppression(pred, conf, iou, agnostic=agnostic, max_det=max_det)
        for b in range(B):
            assert cnt[b] == ref[b].shape[0], (b, cnt[b], ref[b].shape)
            assert torch.equal(rows[b, :cnt[b]], ref[b]), b


def test_nms_argument_errors(dev):
    eng = HipLocalizer(init_yolov5s_state_dict(2, seed=0), input_shape=(64, 64), device=dev)
    with pytest.raises(AssertionError):
        eng.nms(torch.zeros(4, 7), 1.5, 0.5)
    with pytest.raises(AssertionError):
        eng.nms(torch.zeros(4, 7), 0.5, -0.1)
    assert eng.nms(torch.zeros(0, 7), 0.3, 0.5).shape == (0, 6)


def test_efflocalizer_end_to_end(dev, tmp_path):
    """EffLocalizer(model_path, iou_thresh, conf_thresh, ...).run(list) -> list of [n,6] tensors, from a uint8 line image,
    from its file on disk, and from a pre-letterboxed array (the three things the reference's run() accepts), against the
    oracle chain letterbox -> network -> non_max_suppression."""
    from PIL import Image
    sd = _busy_state_dict(2, seed=4)
    path = tmp_path / "yolo_line.pt"
    torch.save(sd, path)
    loc = EffLocalizer(str(path), iou_thresh=0.3, conf_thresh=0.4, num_cores=4, providers=["whatever"], device=dev)
    rng = np.random.default_rng(5)
    im = (rng.integers(0, 256, (48, 400, 3)) // 64 * 64).astype(np.uint8)
    png = tmp_path / "line.png"
    Image.fromarray(im).save(png)
    pre = Y.load_localizer_img(im, (640, 640), bgr=False)
    outs = loc.run([im, str(png), pre]) if False else loc([im, str(png), pre])
    assert isinstance(outs, list) and len(outs) == 3 and all(o.dim() == 2 and o.shape[1] == 6 and o.device.type == "cpu" for o in outs)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = Y.non_max_suppression(Y.yolov5s_forward(sd, torch.from_numpy(pre)), 0.4, 0.3, max_det=1000)[0]
    assert outs[0].shape[0] > 5
    # the device network differs from the CPU one by fp32 summation order, so a box sitting exactly at a threshold may
    # differ: compare as sets with a tolerance
    assert abs(outs[0].shape[0] - ref.shape[0]) <= max(2, ref.shape[0] // 50)
    matched = 0
    for r in ref:
        d = (outs[0][:, :4] - r[:4]).abs().max(1)[0] + (outs[0][:, 5] != r[5]).float() * 1e3
        matched += int(d.min().item() < 0.05)
    assert matched >= ref.shape[0] - max(2, ref.shape[0] // 50)
    bboxes, labels = outs[0][:, :4], outs[0][:, -1]                          # what infer_effocr_onnx_multi.py:255-256 reads
    assert set(labels.tolist()) <= {0.0, 1.0} and (bboxes[:, 2] >= bboxes[:, 0]).all()
    with pytest.raises(NotImplementedError):
        EffLocalizer(sd, model_backend="mmdetection", device=dev)


def test_localizer_feeds_the_recognizer(dev):
    """BASELINE config 5 in miniature: line image -> EffLocalizer -> char boxes scaled back as infer_effocr_onnx_multi.py:313-318
    -> device crops -> encoder -> kNN.  The index holds the embeddings of those very crops, so every box finds itself."""
    from effocr_amd.encoders import HipEncoder
    from effocr_amd.knn import FaissKNN, IndexFlatIP
    from effocr_amd.pipeline import Recognizer
    from effocr_amd.transforms import PairedTransform
    from effocr_amd.weights import init_state_dict
    sd = _busy_state_dict(2, seed=6)
    loc = EffLocalizer(sd, iou_thresh=0.05, conf_thresh=0.5, device=dev)
    rng = np.random.default_rng(7)
    im = (rng.integers(0, 256, (40, 640, 3)) // 32 * 32).astype(np.uint8)
    res = loc([im])[0]
    chars = res[res[:, -1] == 0][:, :4]
    assert chars.shape[0] >= 3
    H, W = im.shape[:2]
    boxes = []
    for bb in chars[:60]:
        x0, _, x1, _ = torch.round(bb)
        x0, x1 = int(round(x0.item() * W / 640)), int(round(x1.item() * W / 640))
        if x1 > x0:
            boxes.append((max(x0, 0), 0, min(x1, W), H))
    esd = init_state_dict("resnet18", seed=1)
    enc = HipEncoder("resnet18", esd, img_size=32, precision="fp32", device=dev)
    tf = PairedTransform(size=32, device=dev)
    emb = enc.forward(tf.boxes(im, boxes, already_int=True), normalize=True)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False, device=dev)
    knn.train(emb)
    rec = Recognizer(enc, knn, [chr(0x4E00 + i) for i in range(len(boxes))], knn=2)
    nearest, _, out = rec.recognize_boxes(im, boxes, char_transform=tf, double_clipped=False)
    uniq = {boxes.index(b) for b in boxes}
    assert len(out) == len(boxes) and sum(nearest[i][0] == chr(0x4E00 + i) for i in uniq) >= len(uniq) - 1
