"""Device-side crop pre-processing: the reference's `create_paired_transform`.

Reference: `utils/datasets_utils.py:69-90` (`MedianPad`), `:166-172` (`create_paired_transform`),
applied per character box at `infer_effocr.py:286-293` / by `TransformationThread`s at
`infer_effocr_onnx_multi.py:326-345`.  There every crop is padded, resized and normalised on the CPU
and the 224x224 fp32 result (602 KB) crosses PCIe; here the uint8 page / line image crosses once and
`effocr_crop_transform` produces the `[n,3,S,S]` encoder input in HBM.

Two call conventions:
  * `create_paired_transform(size)(crop)` — the reference's per-crop callable (HWC uint8 ndarray or PIL
    image -> `Tensor[3,S,S]`), kept so `EffOCR(char_transform=...)` works unchanged;
  * `PairedTransform.boxes(image, bboxes)` — the batched form the drivers should use: one image, the
    localizer's float boxes, rounded (`map(int, map(round, bbox))`) and sliced with numpy semantics.
No CPU fallback: the HIP extension and a ROCm device are required.
"""
import ctypes

import numpy as np
import torch

from . import _lib

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def round_boxes(bboxes):
    """`x0, y0, x1, y1 = map(int, map(round, bbox))` (infer_effocr.py:287) for a sequence of boxes."""
    return [tuple(int(round(float(v))) for v in b[:4]) for b in bboxes]


def slice_boxes(boxes, height, width):
    """Resolve integer boxes the way `im[y0:y1, x0:x1]` does (negative indices wrap, ends clip).
    Raises ValueError for a box that selects nothing — the reference dies there too (PIL cannot
    build a zero-sized image; infer_effocr.py:294-297)."""
    b = np.asarray(list(boxes), np.int64).reshape(-1, 4)

    def resolve(v, size):                                  # slice(start, stop).indices(size) for step 1
        v = np.where(v < 0, v + size, v)
        return np.clip(v, 0, size)
    out = np.stack([resolve(b[:, 0], width), resolve(b[:, 1], height), resolve(b[:, 2], width), resolve(b[:, 3], height)], 1)
    bad = np.nonzero((out[:, 2] <= out[:, 0]) | (out[:, 3] <= out[:, 1]))[0]
    if bad.size:
        i = int(bad[0])
        raise ValueError(f"box {i} = {tuple(int(v) for v in b[i])} selects an empty crop of a {width}x{height} image")
    return np.ascontiguousarray(out.astype(np.int32))


_OUT_PREC = {torch.float32: _lib.PREC["fp32"], torch.float16: _lib.PREC["fp16"], torch.bfloat16: _lib.PREC["bf16"]}


def _f3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


class PairedTransform:
    """`T.Compose([MedianPad(override=fill), T.ToTensor(), T.Resize((size, size)), T.Normalize(mean, std)])`."""

    def __init__(self, size=224, antialias=True, fill=(255, 255, 255), mean=IMAGENET_DEFAULT_MEAN,
                 std=IMAGENET_DEFAULT_STD, device=None):
        if size <= 0 or size % 4:
            raise ValueError("size must be a positive multiple of 4")
        if fill is None:
            raise NotImplementedError("MedianPad(override=None) (median border colour) is a training-time "
                                      "augmentation; inference always overrides with white")
        self.size, self.antialias = int(size), bool(antialias)
        self.fill, self.mean, self.std = tuple(fill), tuple(mean), tuple(std)
        self.device = torch.device(device) if device is not None else None

    def _dev(self):
        self.device = d = _lib.require_gpu(self.device)      # None / "cuda" = the current device
        return d

    def upload(self, image):
        """HWC uint8 ndarray / tensor -> contiguous device tensor (the only PCIe transfer of the path)."""
        if isinstance(image, torch.Tensor):
            t = image
        else:
            arr = np.asarray(image)
            if arr.ndim != 3 or arr.shape[2] != 3 or arr.dtype != np.uint8:
                raise ValueError("image must be HWC uint8 RGB")
            t = torch.from_numpy(np.ascontiguousarray(arr))
        if t.dim() != 3 or t.shape[2] != 3 or t.dtype != torch.uint8:
            raise ValueError("image must be HWC uint8 RGB")
        return t.to(self._dev(), non_blocking=True).contiguous()

    def boxes(self, image, bboxes, out=None, already_int=False):
        """All crops of one image: -> Tensor[n,3,S,S] fp32 on the device.  `bboxes`: iterable of
        (x0,y0,x1,y1[,score]) floats from the localizer (rounded here) or ints with already_int=True."""
        img = self.upload(image)
        H, W = int(img.shape[0]), int(img.shape[1])
        ib = slice_boxes(list(bboxes) if already_int else round_boxes(bboxes), H, W)
        n, S = len(ib), self.size
        dev = img.device
        if out is None:
            out = torch.empty((n, 3, S, S), dtype=torch.float32, device=dev)
        elif tuple(out.shape) != (n, 3, S, S) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
            raise ValueError("out must be a contiguous float32 [n,3,size,size] tensor on the image's device")
        if n == 0:
            return out
        L = _lib.lib()
        for lo in range(0, n, 65535):
            hi = min(n, lo + 65535)
            bx = torch.from_numpy(ib[lo:hi]).to(dev)
            with torch.cuda.device(dev):
                _lib.check(L.effocr_crop_transform(_lib.ptr(img), H, W, W * 3, _lib.ptr(bx), hi - lo, S, int(self.antialias),
                                                   _f3(self.mean), _f3(self.std), _f3(self.fill), _lib.ptr(out[lo:hi]),
                                                   _lib.current_stream(dev)), "crop_transform")
            # bx is freed by the caching allocator in stream order: safe without a sync
        return out

    def boxes_batch(self, images, boxes5, out=None, dtype=torch.float32):
        """The crops of SEVERAL images of one geometry in one launch (effocr_crop_transform_batch): ``images`` = uint8 device tensor
        [L,H,W,3] (contiguous), ``boxes5`` = int32 device tensor [n,5] = x0,y0,x1,y1,image index, coordinates already resolved the
        way numpy slicing does (``slice_boxes`` semantics).  An EMPTY box gives a zero crop — what the reference's
        ``create_batches`` substitutes for a crop whose transform raised (infer_effocr_onnx_multi.py:145-147,196-200).
        ``dtype``: torch.float32 (the reference's crop type) or torch.float16 / torch.bfloat16 — SURVEY f-2's 16-bit hand-off: the
        fp32 result rounded once to the encoder's operand type (``HipEncoder.crop_dtype``), which the patch embedding would do
        itself; same embeddings bit for bit, half the bytes written here and read there.
        -> Tensor[n,3,S,S] on the device, nothing synchronised."""
        if dtype not in _OUT_PREC:
            raise ValueError("dtype must be torch.float32, torch.float16 or torch.bfloat16")
        if images.dim() != 4 or images.shape[3] != 3 or images.dtype != torch.uint8 or not images.is_cuda:
            raise ValueError("images must be a uint8 device tensor [L,H,W,3]")
        if boxes5.dim() != 2 or boxes5.shape[1] != 5 or boxes5.dtype != torch.int32 or boxes5.device != images.device:
            raise ValueError("boxes5 must be an int32 [n,5] tensor on the images' device")
        images, boxes5 = images.contiguous(), boxes5.contiguous()
        Lc, H, W = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
        n, S, dev = int(boxes5.shape[0]), self.size, images.device
        if out is None:
            out = torch.empty((n, 3, S, S), dtype=dtype, device=dev)
        elif tuple(out.shape) != (n, 3, S, S) or out.dtype != dtype or not out.is_contiguous() or out.device != dev:
            raise ValueError(f"out must be a contiguous {dtype} [n,3,size,size] tensor on the images' device")
        if n == 0:
            return out
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().effocr_crop_transform_batch_ex(_lib.ptr(images), Lc, H * W * 3, H, W, W * 3, _lib.ptr(boxes5), n, S,
                                                                 int(self.antialias), _f3(self.mean), _f3(self.std), _f3(self.fill),
                                                                 _OUT_PREC[dtype], _lib.ptr(out), _lib.current_stream(dev)),
                       "crop_transform_batch")
        return out

    def __call__(self, crop):
        """Reference per-crop convention: one HWC uint8 crop -> Tensor[3,S,S] (on the device)."""
        arr = np.asarray(crop)
        if arr.ndim != 3 or arr.shape[0] == 0 or arr.shape[1] == 0:
            raise ValueError("empty crop")
        h, w = arr.shape[:2]
        return self.boxes(arr, [(0, 0, w, h)], already_int=True)[0]


def create_paired_transform(size=224, antialias=True, device=None):
    """utils/datasets_utils.py:166-172."""
    return PairedTransform(size=size, antialias=antialias, device=device)
