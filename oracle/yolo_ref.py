"""TEST INFRASTRUCTURE — CPU restatement of the YOLOv5 localizer path (never imported by effocr_amd/).

PARITY UNPINNED: the reference runs an exported ``.onnx`` through ONNXRuntime (onnx_engines/localizer_engine.py:24-28,52)
and ships neither the model definition nor weights; ultralytics, cv2, torchvision and onnxruntime are not installable
here.  This file restates
  * the network from the published ultralytics YOLOv5 v6 sources (models/yolov5s.yaml; models/common.py Conv / Bottleneck /
    C3 / SPPF; models/yolo.py Detect inference branch) as literal ``torch.nn.functional`` calls over the state dict,
  * ``EffLocalizer.letterbox`` / ``load_localizer_img`` (localizer_engine.py:75-85,107-138) with cv2.resize(INTER_LINEAR)'s
    uint8 fixed-point arithmetic restated from OpenCV's resize.cpp (HResizeLinear / VResizeLinear, 11-bit coefficients),
  * ``EffLocalizer.non_max_suppression`` (:171-277) line by line for the branch the engine uses (single label, no masks,
    classes=None), with ``torchvision.ops.nms`` restated as the greedy loop of its documentation, and the one thing the
    reference leaves open DEFINED: equal confidences keep ascending row order (its ``argsort(descending=True)`` is unstable).
"""
import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------- network
def _conv(sd, name, x, k, s):
    """ultralytics Conv: Conv2d(bias=False, padding=k//2) + BatchNorm2d(eps=1e-3) + SiLU."""
    x = F.conv2d(x, sd[name + ".conv.weight"], None, stride=s, padding=k // 2 if k != 6 else 2)
    x = F.batch_norm(x, sd[name + ".bn.running_mean"], sd[name + ".bn.running_var"], sd[name + ".bn.weight"], sd[name + ".bn.bias"],
                     training=False, eps=1e-3)
    return F.silu(x)


def _c3(sd, name, x, n, shortcut):
    a = _conv(sd, name + ".cv1", x, 1, 1)
    for i in range(n):
        y = _conv(sd, f"{name}.m.{i}.cv2", _conv(sd, f"{name}.m.{i}.cv1", a, 1, 1), 3, 1)
        a = a + y if shortcut else y
    return _conv(sd, name + ".cv3", torch.cat((a, _conv(sd, name + ".cv2", x, 1, 1)), 1), 1, 1)


def _sppf(sd, name, x):
    x = _conv(sd, name + ".cv1", x, 1, 1)
    y1 = F.max_pool2d(x, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    return _conv(sd, name + ".cv2", torch.cat((x, y1, y2, F.max_pool2d(y2, 5, 1, 2)), 1), 1, 1)


def yolov5s_forward(sd, x):
    """x [B,3,H,W] fp32 (letterboxed, 0..1) -> [B, sum(3*ny*nx), 5+nc]: output 0 of the exported model."""
    sd = {k: v.float() for k, v in sd.items()}
    x0 = _conv(sd, "model.0", x, 6, 2)
    x1 = _conv(sd, "model.1", x0, 3, 2)
    x2 = _c3(sd, "model.2", x1, 1, True)
    x3 = _conv(sd, "model.3", x2, 3, 2)
    x4 = _c3(sd, "model.4", x3, 2, True)
    x5 = _conv(sd, "model.5", x4, 3, 2)
    x6 = _c3(sd, "model.6", x5, 3, True)
    x7 = _conv(sd, "model.7", x6, 3, 2)
    x8 = _c3(sd, "model.8", x7, 1, True)
    x9 = _sppf(sd, "model.9", x8)
    x10 = _conv(sd, "model.10", x9, 1, 1)
    x12 = torch.cat((F.interpolate(x10, scale_factor=2, mode="nearest"), x6), 1)
    x13 = _c3(sd, "model.13", x12, 1, False)
    x14 = _conv(sd, "model.14", x13, 1, 1)
    x16 = torch.cat((F.interpolate(x14, scale_factor=2, mode="nearest"), x4), 1)
    x17 = _c3(sd, "model.17", x16, 1, False)
    x19 = torch.cat((_conv(sd, "model.18", x17, 3, 2), x14), 1)
    x20 = _c3(sd, "model.20", x19, 1, False)
    x22 = torch.cat((_conv(sd, "model.21", x20, 3, 2), x10), 1)
    x23 = _c3(sd, "model.23", x22, 1, False)
    # Detect (inference): per level conv -> (bs, na, no, ny, nx) -> (bs, na, ny, nx, no) -> sigmoid -> decode -> (bs, na*ny*nx, no)
    anchors = sd["model.24.anchors"]                       # [3,3,2] in stride units
    strides = (8.0, 16.0, 32.0)
    z = []
    for l, f in enumerate((x17, x20, x23)):
        r = F.conv2d(f, sd[f"model.24.m.{l}.weight"], sd[f"model.24.m.{l}.bias"])
        bs, _, ny, nx = r.shape
        no = r.shape[1] // 3
        y = r.view(bs, 3, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous().sigmoid()
        yv, xv = torch.meshgrid(torch.arange(ny, dtype=torch.float32), torch.arange(nx, dtype=torch.float32), indexing="ij")
        grid = torch.stack((xv, yv), 2).expand(1, 3, ny, nx, 2) - 0.5
        anchor_grid = (anchors[l] * strides[l]).view(1, 3, 1, 1, 2).expand(1, 3, ny, nx, 2)
        xy = (y[..., 0:2] * 2 + grid) * strides[l]
        wh = (y[..., 2:4] * 2) ** 2 * anchor_grid
        z.append(torch.cat((xy, wh, y[..., 4:]), 4).view(bs, 3 * ny * nx, no))
    return torch.cat(z, 1)


# ------------------------------------------------------------------------------------------------- letterbox
def _cv_round(v):
    return np.rint(v).astype(np.int64)                     # cvRound: round half to even


def _lin_coef(dst, src):
    scale = src / dst
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, _cv_round((np.float32(1.0) - f) * np.float32(2048.0)), _cv_round(f * np.float32(2048.0))


def resize_linear_u8(im, new_w, new_h):
    """cv2.resize(im, (new_w, new_h), interpolation=cv2.INTER_LINEAR) for HWC uint8 (fixed-point path)."""
    H, W = im.shape[:2]
    x0, x1, ax0, ax1 = _lin_coef(new_w, W)
    y0, y1, by0, by1 = _lin_coef(new_h, H)
    src = im.astype(np.int64)
    S = src[:, x0, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]          # [H, new_w, 3], x 2^11
    S0, S1 = S[y0], S[y1]
    r = (((by0[:, None, None] * (S0 >> 4)) >> 16) + ((by1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(r, 0, 255).astype(np.uint8)


def letterbox(im, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True, stride=32):
    """localizer_engine.py:107-138."""
    shape = im.shape[:2]
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    elif scaleFill:
        dw, dh = 0.0, 0.0
        new_unpad = (new_shape[1], new_shape[0])
        ratio = new_shape[1] / shape[1], new_shape[0] / shape[0]
    dw /= 2
    dh /= 2
    if shape[::-1] != new_unpad:
        im = resize_linear_u8(im, new_unpad[0], new_unpad[1])
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.empty((im.shape[0] + top + bottom, im.shape[1] + left + right, 3), dtype=np.uint8)
    out[...] = np.array(color, dtype=np.uint8)
    out[top:top + im.shape[0], left:left + im.shape[1]] = im
    return out, ratio, (dw, dh)


def load_localizer_img(im0, input_shape, bgr=True):
    """localizer_engine.py:75-85 from the decoded image on (im0 = cv2.imread result, BGR)."""
    im = letterbox(im0, input_shape, stride=32, auto=False)[0]
    im = im.transpose((2, 0, 1))
    if bgr:
        im = im[::-1]
    im = np.ascontiguousarray(im).astype(np.float32) / 255.0
    return np.expand_dims(im, 0)


# ------------------------------------------------------------------------------------------------- NMS
def xywh2xyxy(x):
    y = x.clone()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def nms_loop(boxes, scores, iou_thres):
    """torchvision.ops.nms: boxes are visited in decreasing score order; a box is kept unless it overlaps an already kept
    box with IoU > iou_thres.  ``scores`` must already be sorted descending (the caller sorts, as the reference does)."""
    keep = []
    bt = boxes
    areas = (bt[:, 2] - bt[:, 0]) * (bt[:, 3] - bt[:, 1])
    for i in range(bt.shape[0]):
        if keep:
            k = torch.tensor(keep)
            iw = (torch.minimum(bt[k, 2], bt[i, 2]) - torch.maximum(bt[k, 0], bt[i, 0])).clamp(min=0)
            ih = (torch.minimum(bt[k, 3], bt[i, 3]) - torch.maximum(bt[k, 1], bt[i, 1])).clamp(min=0)
            inter = iw * ih
            iou = inter / (areas[k] + areas[i] - inter)
            if bool((iou > iou_thres).any()):
                continue
        keep.append(i)
    return torch.tensor(keep, dtype=torch.int64)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, agnostic=False, max_det=300):
    """localizer_engine.py:171-277, single-label / no-mask / classes=None branch.  prediction [bs, n, 5+nc] -> list of [m,6]."""
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1
    prediction = prediction.clone().float()
    bs = prediction.shape[0]
    xc = prediction[..., 4] > conf_thres
    max_wh, max_nms = 7680, 30000
    output = [torch.zeros((0, 6))] * bs
    for xi, x in enumerate(prediction):
        x = x[xc[xi]]
        if not x.shape[0]:
            continue
        x[:, 5:] *= x[:, 4:5]
        box = xywh2xyxy(x[:, :4])
        conf, j = x[:, 5:].max(1, keepdim=True)
        x = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]
        n = x.shape[0]
        if not n:
            continue
        order = torch.sort(x[:, 4], descending=True, stable=True)[1]          # DEFINED: ties keep ascending row order
        x = x[order[:max_nms]]
        c = x[:, 5:6] * (0 if agnostic else max_wh)
        boxes, scores = x[:, :4] + c, x[:, 4]
        i = nms_loop(boxes, scores, iou_thres)
        if i.shape[0] > max_det:
            i = i[:max_det]
        output[xi] = x[i]
    return output
