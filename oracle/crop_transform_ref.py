"""CPU restatement of the reference's per-crop pre-processing (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

Parity unpinned: the reference holds no test or fixture for this step, and torchvision / PIL-based
`utils/datasets_utils.py` cannot be imported here (torchvision, timm, albumentations, kornia absent).

What it follows:
  * crop                 `im[y0:y1, x0:x1, :]` on the HWC uint8 page/line image, coordinates
                         `map(int, map(round, bbox))`                      infer_effocr.py:286-293
  * `MedianPad(override=(255,255,255))`: pad RIGHT and BOTTOM to a square of side max(h, w) with the
    override colour (`T.Pad((0, 0, pad_x, pad_y), fill=...)`)              utils/datasets_utils.py:69-90
  * `T.ToTensor()`: HWC uint8 -> CHW float32 / 255
  * `T.Resize((size, size))` on a float tensor = `torch.nn.functional.interpolate(mode="bilinear",
    align_corners=False, antialias=A)`; A is torchvision-version dependent (True from 0.17 on, False
    before) -> both are restated, `antialias=True` is the default here
  * `T.Normalize(IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD)`            utils/datasets_utils.py:166-172

`resize_weights` restates ATen's index/weight computation (UpSampleKernel.cpp, `HelperInterpLinear`):
  antialias: centre = scale*(i+0.5), support = max(scale, 1), triangle filter, weights normalised;
  plain:     src = max(scale*(i+0.5)-0.5, 0), two taps (i0, min(i0+1, n-1)) with weights (1-l, l).
The separable application order is width first, then height, with an fp32 intermediate.
"""
import numpy as np

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def python_slice_box(box, height, width):
    """(x0, y0, x1, y1) of ints -> the bounds numpy basic slicing `im[y0:y1, x0:x1]` actually uses."""
    x0, y0, x1, y1 = (int(v) for v in box)
    xs, xe, _ = slice(x0, x1).indices(width)
    ys, ye, _ = slice(y0, y1).indices(height)
    return xs, ys, max(xe, xs), max(ye, ys)


def round_box(bbox):
    """`map(int, map(round, bbox))` (Python round = half to even)."""
    return tuple(int(round(float(v))) for v in bbox)


def resize_weights(in_size, out_size, antialias):
    """-> (start[out] int, weights[out, T] float32 zero padded, count[out])."""
    scale = np.float32(in_size) / np.float32(out_size)
    if antialias:
        support = np.float32(scale) if scale >= 1.0 else np.float32(1.0)
        invscale = np.float32(1.0) / scale if scale >= 1.0 else np.float32(1.0)
        T = int(np.ceil(support)) * 2 + 1
        start = np.zeros(out_size, np.int64)
        count = np.zeros(out_size, np.int64)
        w = np.zeros((out_size, T), np.float32)
        for i in range(out_size):
            center = np.float32(scale * np.float32(i + 0.5))
            xmin = max(int(np.float32(center - support + np.float32(0.5))), 0)
            xsize = min(int(np.float32(center + support + np.float32(0.5))), in_size) - xmin
            xsize = min(max(xsize, 0), T)
            tot = np.float32(0)
            for j in range(xsize):
                v = np.float32(1.0) - abs(np.float32((np.float32(j + xmin) - center + np.float32(0.5)) * invscale))
                v = v if v > 0 else np.float32(0)
                w[i, j] = v
                tot += v
            if tot != 0:
                w[i, :xsize] /= tot
            start[i], count[i] = xmin, xsize
        return start, w, count
    start = np.zeros(out_size, np.int64)
    w = np.zeros((out_size, 2), np.float32)
    for i in range(out_size):
        src = np.float32(scale * np.float32(i + 0.5) - np.float32(0.5))
        src = src if src > 0 else np.float32(0)
        i0 = min(int(src), in_size - 1)
        lam = min(max(np.float32(src - np.float32(i0)), np.float32(0)), np.float32(1))
        start[i] = i0
        if i0 + 1 < in_size:
            w[i, 0], w[i, 1] = np.float32(1) - lam, lam
        else:                                  # i1 == i0: both taps hit the last pixel
            w[i, 0], w[i, 1] = np.float32(1), np.float32(0)
    return start, w, np.full(out_size, 2, np.int64)


def pad_square_to_float(crop, fill=(255, 255, 255)):
    """HWC uint8 crop -> CHW float32 in [0,1], padded right/bottom to a square with `fill`."""
    h, w, c = crop.shape
    if h == 0 or w == 0:
        raise ValueError("empty crop")         # PIL raises on a zero-sized image (infer_effocr.py:294-297)
    L = max(h, w)
    sq = np.empty((L, L, c), np.uint8)
    sq[...] = np.asarray(fill, np.uint8)
    sq[:h, :w] = crop
    return np.ascontiguousarray(sq.transpose(2, 0, 1)).astype(np.float32) / np.float32(255)


def resize_bilinear(x, size, antialias=True):
    """CHW float32 -> [C, size, size]; separable, width first then height (numpy restatement)."""
    C, H, W = x.shape
    sx, wx, cx = resize_weights(W, size, antialias)
    sy, wy, cy = resize_weights(H, size, antialias)
    tmp = np.zeros((C, H, size), np.float32)
    for i in range(size):
        acc = np.zeros((C, H), np.float32)
        for j in range(int(cx[i])):
            acc += wx[i, j] * x[:, :, min(int(sx[i]) + j, W - 1)]
        tmp[:, :, i] = acc
    out = np.zeros((C, size, size), np.float32)
    for i in range(size):
        acc = np.zeros((C, size), np.float32)
        for j in range(int(cy[i])):
            acc += wy[i, j] * tmp[:, min(int(sy[i]) + j, H - 1), :]
        out[:, i, :] = acc
    return out


def resize_bilinear_torch(x, size, antialias=True):
    """Same through torch (what torchvision's tensor Resize dispatches to)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x))[None]
    return torch.nn.functional.interpolate(t, size=(size, size), mode="bilinear", align_corners=False,
                                           antialias=bool(antialias))[0].numpy()


def normalize(x, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD):
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return (x - m) / s


def paired_transform(crop, size=224, antialias=True, use_torch=True):
    """`create_paired_transform(size)(crop)`: HWC uint8 -> [3, size, size] float32."""
    x = pad_square_to_float(np.asarray(crop))
    y = resize_bilinear_torch(x, size, antialias) if use_torch else resize_bilinear(x, size, antialias)
    return normalize(y).astype(np.float32)


def transform_boxes(image, boxes, size=224, antialias=True, use_torch=True):
    """Batch form of infer_effocr.py:286-293: one HWC uint8 image + integer boxes -> [n, 3, size, size]."""
    H, W = image.shape[:2]
    out = []
    for b in boxes:
        x0, y0, x1, y1 = python_slice_box(b, H, W)
        out.append(paired_transform(image[y0:y1, x0:x1, :], size, antialias, use_torch))
    return np.stack(out) if out else np.zeros((0, 3, size, size), np.float32)
