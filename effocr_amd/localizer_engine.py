"""``EffLocalizer`` — drop-in for onnx_engines/localizer_engine.py:14-66 (``model_backend='yolo'``) on MI355X.

The reference wraps an ONNXRuntime session over an exported ultralytics YOLOv5 model: ``run(imgs)`` letterboxes each
image to ``input_shape`` (:75-85,107-138), runs the network, and ``_postprocess`` (:54-60) applies
``non_max_suppression(pred, conf_thres, iou_thres, max_det=1000)[0]`` to every image, returning a list of ``[n,6]``
tensors ``(x1, y1, x2, y2, conf, cls)`` in letterboxed-input pixels.  The driver reads ``result[:, :4]`` and ``result[:, -1]``
(infer_effocr_onnx_multi.py:252-256: class 0 = characters, class 1 = words).

Here ``model_path`` is a torch state dict (``.pt`` / ``.pth`` / ``.safetensors``, or a dict) with the ultralytics YOLOv5s keys
(``model.0.conv.weight`` ... ``model.24.m.2.bias``, ``model.24.anchors``) instead of an ``.onnx`` graph; everything from the
uint8 image to the kept boxes runs on the device (letterbox kernel -> fp32-MFMA convolutions -> decode -> NMS kernels):
one uint8 upload and one tiny ``[n,6]`` download per image.  Only the ``yolo`` backend exists (the detectron2 / mmdetection
branches of the reference return raw session outputs and are out of scope).
"""
import ctypes
import math
import threading

import numpy as np
import torch

from . import _lib

MAX_WH = 7680.0          # localizer_engine.py:205
MAX_NMS = 30000          # :206
YOLOV5_ANCHORS = ((10, 13, 16, 30, 33, 23), (30, 61, 62, 45, 59, 119), (116, 90, 156, 198, 373, 326))   # pixels, P3 / P4 / P5
STRIDES = (8.0, 16.0, 32.0)


def letterbox_geometry(shape, new_shape=(640, 640), auto=False, scaleFill=False, scaleup=True, stride=32):
    """The arithmetic of ``EffLocalizer.letterbox`` (localizer_engine.py:107-138) without the pixels:
    (h, w) -> (new_h, new_w, top, bottom, left, right, ratio, (dw, dh))."""
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
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_unpad[1], new_unpad[0], top, bottom, left, right, ratio, (dw, dh)


def yolov5s_param_shapes(nc):
    """{ultralytics key: shape} of YOLOv5s (v6 yaml, width 0.5 / depth 0.33) with ``nc`` classes."""
    shapes = {}

    def conv(name, c1, c2, k):
        shapes[name + ".conv.weight"] = (c2, c1, k, k)
        for s in ("weight", "bias", "running_mean", "running_var"):
            shapes[f"{name}.bn.{s}"] = (c2,)

    def c3(name, c1, c2, n):
        c_ = c2 // 2
        conv(name + ".cv1", c1, c_, 1)
        conv(name + ".cv2", c1, c_, 1)
        conv(name + ".cv3", 2 * c_, c2, 1)
        for i in range(n):
            conv(f"{name}.m.{i}.cv1", c_, c_, 1)
            conv(f"{name}.m.{i}.cv2", c_, c_, 3)

    conv("model.0", 3, 32, 6)
    conv("model.1", 32, 64, 3)
    c3("model.2", 64, 64, 1)
    conv("model.3", 64, 128, 3)
    c3("model.4", 128, 128, 2)
    conv("model.5", 128, 256, 3)
    c3("model.6", 256, 256, 3)
    conv("model.7", 256, 512, 3)
    c3("model.8", 512, 512, 1)
    conv("model.9.cv1", 512, 256, 1)
    conv("model.9.cv2", 1024, 512, 1)
    conv("model.10", 512, 256, 1)
    c3("model.13", 512, 256, 1)
    conv("model.14", 256, 128, 1)
    c3("model.17", 256, 128, 1)
    conv("model.18", 128, 128, 3)
    c3("model.20", 256, 256, 1)
    conv("model.21", 256, 256, 3)
    c3("model.23", 512, 512, 1)
    for l, c in enumerate((128, 256, 512)):
        shapes[f"model.24.m.{l}.weight"] = (3 * (nc + 5), c, 1, 1)
        shapes[f"model.24.m.{l}.bias"] = (3 * (nc + 5),)
    shapes["model.24.anchors"] = (3, 3, 2)
    return shapes


def init_yolov5s_state_dict(nc=2, seed=0):
    """Seeded random YOLOv5s weights (there is no network to fetch a trained localizer): kaiming-uniform convolutions,
    BatchNorm with non-trivial statistics, ultralytics' Detect bias initialisation, the default COCO anchors."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in yolov5s_param_shapes(nc).items():
        if k.endswith("conv.weight") or (k.startswith("model.24.m.") and k.endswith(".weight")):
            fan_in = shp[1] * shp[2] * shp[3]
            bound = math.sqrt(3.0 / fan_in) * 1.2
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound
        elif k.endswith("bn.weight"):
            sd[k] = 1.0 + 0.2 * torch.randn(shp, generator=g)
        elif k.endswith("bn.bias") or k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith(".bias"):                              # Detect._initialize_biases: obj ~ 8 objects / 640 image, cls ~ 0.6 / nc
            l = int(k.split(".")[3])
            b = 0.05 * torch.randn(3, nc + 5, generator=g)
            b[:, 4] += math.log(8 / (640 / STRIDES[l]) ** 2)
            b[:, 5:] += math.log(0.6 / (nc - 0.99999))
            sd[k] = b.reshape(-1)
        elif k == "model.24.anchors":
            sd[k] = torch.tensor(YOLOV5_ANCHORS, dtype=torch.float32).view(3, 3, 2) / torch.tensor(STRIDES).view(3, 1, 1)
    return sd


def _load_state_dict(model_path):
    if isinstance(model_path, dict):
        sd = model_path
    elif str(model_path).endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(str(model_path))
    else:
        sd = torch.load(str(model_path), map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
            sd = sd["state_dict"]
    return {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}


class HipLocalizer:
    """Device-resident YOLOv5s: C-ABI handle + weight blob + per-stream workspaces."""

    def __init__(self, state_dict, input_shape=(640, 640), device=None, precision="fp32"):
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
        self.precision = precision
        self.device = _lib.require_gpu(device)       # None / "cuda" = the current device
        self._L = _lib.lib()
        key = "model.24.m.0.bias"
        if key not in state_dict:
            raise ValueError(f"state dict has no '{key}': not an ultralytics YOLOv5 detection model")
        no = int(state_dict[key].numel()) // 3
        self.nc, self.no = no - 5, no
        self.input_shape = (int(input_shape[0]), int(input_shape[1]))
        want = yolov5s_param_shapes(self.nc)
        bad = [f"missing {k}" for k in want if k not in state_dict] + \
              [f"{k}: shape {tuple(state_dict[k].shape)} != {want[k]}" for k in want if k in state_dict and tuple(state_dict[k].shape) != tuple(want[k])]
        if bad:
            raise ValueError("state dict does not match yolov5s: " + "; ".join(bad[:6]) + (f" (+{len(bad) - 6} more)" if len(bad) > 6 else ""))
        self._h = ctypes.c_void_p()
        _lib.check(self._L.effocr_localizer_create(b"yolov5s", self.nc, self.input_shape[0], self.input_shape[1], ctypes.byref(self._h)),
                   "effocr_localizer_create", self._L)
        for i in range(self._L.effocr_localizer_num_params(self._h)):
            name = self._L.effocr_localizer_param_name(self._h, i).decode()
            t = state_dict[name].detach().to("cpu", torch.float32).contiguous()
            _lib.check(self._L.effocr_localizer_set_param(self._h, name.encode(), _lib.ptr(t), t.numel()), f"effocr_localizer_set_param({name})", self._L)
        nbytes = int(self._L.effocr_localizer_weights_bytes(self._h))
        with torch.cuda.device(self.device):
            self._wblob = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(self._L.effocr_localizer_upload(self._h, _lib.ptr(self._wblob), nbytes), "effocr_localizer_upload", self._L)
        self.set_option("bf16_operands", 1 if precision == "bf16" else 0)
        self.num_predictions = int(self._L.effocr_localizer_num_predictions(self._h))
        self._ws = {}
        self._lock = threading.Lock()

    def set_option(self, name, value):
        _lib.check(self._L.effocr_localizer_set_option(self._h, name.encode(), int(value)), "effocr_localizer_set_option", self._L)

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self._L.effocr_localizer_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    def _workspace(self, tag, need):
        key = (tag, torch.cuda.current_stream(self.device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(int(need), 256), dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def forward(self, x):
        """x: [B,3,H,W] float32 CUDA tensor (letterboxed, RGB, 0..1) -> [B, num_predictions, 5 + nc] (async)."""
        if x.dim() != 4 or x.shape[1] != 3 or tuple(x.shape[2:]) != self.input_shape:
            raise ValueError(f"expected input [B,3,{self.input_shape[0]},{self.input_shape[1]}], got {tuple(x.shape)}")
        if x.dtype != torch.float32 or x.device != self.device:
            raise ValueError(f"expected a float32 tensor on {self.device}, got {x.dtype} on {x.device}")
        x = x.contiguous()
        B = x.shape[0]
        pred = torch.empty((B, self.num_predictions, self.no), dtype=torch.float32, device=self.device)
        if B == 0:
            return pred
        need = int(self._L.effocr_localizer_workspace_bytes(self._h, B))
        with self._lock, torch.cuda.device(self.device):
            ws = self._workspace("fwd", need)
            _lib.check(self._L.effocr_localizer_forward(self._h, _lib.ptr(x), B, _lib.ptr(pred), _lib.ptr(ws), ws.numel(),
                                                        _lib.current_stream(self.device)), "effocr_localizer_forward", self._L)
        return pred

    def letterbox(self, image, bgr=False, out=None):
        """HWC uint8 image (numpy / tensor; ``bgr=True`` for cv2.imread order) -> [1,3,H,W] float32 on the device
        (load_localizer_img, localizer_engine.py:75-85).  ``out``: an existing [1,3,H,W] slice of a batch tensor to fill."""
        t = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image))
        if t.dim() != 3 or t.shape[2] != 3 or t.dtype != torch.uint8:
            raise ValueError("image must be HWC uint8 with 3 channels")
        t = t.to(self.device).contiguous()
        H, W = int(t.shape[0]), int(t.shape[1])
        nh, nw, top, _, left, _, _, _ = letterbox_geometry((H, W), self.input_shape, auto=False, stride=32)
        if out is None:
            out = torch.empty((1, 3) + self.input_shape, dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != (1, 3) + self.input_shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous float32 [1,3,H,W] tensor on the localizer's device")
        with torch.cuda.device(self.device):
            _lib.check(self._L.effocr_letterbox(_lib.ptr(t), H, W, 3 * W, 1 if bgr else 0, self.input_shape[0], self.input_shape[1], nh, nw, top, left,
                                                _lib.ptr(out), _lib.current_stream(self.device)), "effocr_letterbox", self._L)
        return out

    def nms_async(self, pred, conf_thres, iou_thres, max_det=1000, agnostic=False, out=None, cnt=None):
        """pred [n, 5 + nc] (one image) -> (rows [max_det,6], count [1] int32), both on the device, no synchronisation.
        ``out`` / ``cnt``: rows of a caller's batch result to fill instead of fresh tensors (cnt must be zeroed by the caller)."""
        if not (0 <= conf_thres <= 1):
            raise AssertionError(f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0")
        if not (0 <= iou_thres <= 1):
            raise AssertionError(f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0")
        pred = pred.to(self.device, torch.float32).contiguous()
        n = int(pred.shape[0])
        if out is None:
            out = torch.empty((max_det, 6), dtype=torch.float32, device=self.device)
        if cnt is None:
            cnt = torch.zeros(1, dtype=torch.int32, device=self.device)
        need = int(self._L.effocr_nms_workspace_bytes(n, MAX_NMS))
        with self._lock, torch.cuda.device(self.device):
            # ONE scratch per stream, reused by every image: the launches of successive images are ordered on the stream, so the
            # candidate lists of image i are dead when image i+1's kernels start (up to ~80 MB at 25 200 candidates: not per image)
            ws = self._workspace("nms", need)
            _lib.check(self._L.effocr_nms(_lib.ptr(pred), n, self.nc, float(conf_thres), float(iou_thres), int(max_det), MAX_NMS, MAX_WH,
                                          1 if agnostic else 0, _lib.ptr(out), _lib.ptr(cnt), _lib.ptr(ws), ws.numel(),
                                          _lib.current_stream(self.device)), "effocr_nms", self._L)
        return out, cnt

    def nms_batch_async(self, pred, conf_thres, iou_thres, max_det=1000, agnostic=False, out=None, cnt=None):
        """pred [B, n, 5 + nc] (the images of one network call) -> (rows [B, max_det, 6], counts [B] int32) on the device, no
        synchronisation: effocr_nms_batch — one launch for all images when n <= 25600 (any max_det; no workspace), else the
        per-image kernels back to back.  ``out`` / ``cnt``: slices of a caller's result to fill."""
        if not (0 <= conf_thres <= 1):
            raise AssertionError(f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0")
        if not (0 <= iou_thres <= 1):
            raise AssertionError(f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0")
        pred = pred.to(self.device, torch.float32).contiguous()
        B, n = int(pred.shape[0]), int(pred.shape[1])
        if out is None:
            out = torch.empty((B, max_det, 6), dtype=torch.float32, device=self.device)
        if cnt is None:
            cnt = torch.zeros(B, dtype=torch.int32, device=self.device)
        assert out.is_contiguous() and cnt.is_contiguous() and out.shape == (B, max_det, 6) and cnt.shape == (B,)
        need = int(self._L.effocr_nms_batch_workspace_bytes(n, int(max_det), MAX_NMS))      # 0: the one-launch greedy path
        with self._lock, torch.cuda.device(self.device):
            ws = self._workspace("nms", max(need, 256))
            _lib.check(self._L.effocr_nms_batch(_lib.ptr(pred), B, n, self.nc, float(conf_thres), float(iou_thres), int(max_det), MAX_NMS, MAX_WH,
                                                1 if agnostic else 0, _lib.ptr(out), _lib.ptr(cnt), _lib.ptr(ws), ws.numel(),
                                                _lib.current_stream(self.device)), "effocr_nms_batch", self._L)
        return out, cnt

    def nms(self, pred, conf_thres, iou_thres, max_det=1000, agnostic=False):
        """pred [n, 5 + nc] (one image) -> [m,6] device tensor, m <= max_det (non_max_suppression(...)[0])."""
        out, cnt = self.nms_async(pred, conf_thres, iou_thres, max_det, agnostic)
        return out[: int(cnt.item())]


class EffLocalizer:

    def __init__(self, model_path, iou_thresh=0.01, conf_thresh=0.30, vertical=False, num_cores=None, providers=None,
                 input_shape=(640, 640), model_backend='yolo', device=None, precision="fp32"):
        # precision (extension): "fp32" = fp32 MFMA operands (the oracle's arithmetic), "bf16" = bf16-rounded operands for every
        # convolution with an activation, fp32 accumulation and fp32 Detect heads (2-3x the network throughput)
        # num_cores / providers are ORT knobs (localizer_engine.py:17-23): accepted and ignored.
        if model_backend != 'yolo':
            raise NotImplementedError('Backend {} is not implemented'.format(model_backend))
        self.num_cores, self.providers = num_cores, providers
        self._iou_thresh, self._conf_thresh, self._vertical = iou_thresh, conf_thresh, vertical
        self._input_shape = (int(input_shape[0]), int(input_shape[1]))
        self._model_backend = model_backend
        self._eng_net = HipLocalizer(_load_state_dict(model_path), input_shape=self._input_shape, device=device, precision=precision)

    def __call__(self, imgs):
        return self.run(imgs)

    def load_localizer_img(self, input_path):
        """localizer_engine.py:75-85 with PIL instead of cv2.imread (cv2 is not installed): RGB in, so no channel swap."""
        from PIL import Image
        im0 = np.array(Image.open(input_path).convert("RGB"))
        return self._eng_net.letterbox(im0, bgr=False)

    def _letterboxed(self, imgs):
        """list entries -> one [n,3,H,W] float32 device tensor (one image per ENTRY).  A pre-letterboxed float entry may carry a
        batch: like the reference, whose ``_postprocess`` keeps ``non_max_suppression(pred)[0]`` per entry
        (localizer_engine.py:58-60), only its first image is used, so results always line up with the input list."""
        eng = self._eng_net
        x = torch.empty((len(imgs), 3) + self._input_shape, dtype=torch.float32, device=eng.device)
        for i, img in enumerate(imgs):
            if isinstance(img, str):
                from PIL import Image
                img = np.array(Image.open(img).convert("RGB"))    # load_localizer_img (:75-85), PIL instead of cv2: RGB in, no channel swap
            if (isinstance(img, np.ndarray) and img.dtype == np.uint8) or (isinstance(img, torch.Tensor) and img.dtype == torch.uint8):
                eng.letterbox(img, bgr=False, out=x[i:i + 1])
            else:
                t = torch.as_tensor(img)
                if t.dtype != torch.float32:
                    raise ValueError(f"Unexpected input data type. Actual: {t.dtype}, expected: float32")
                if t.dim() == 3:
                    t = t.unsqueeze(0)
                if t.dim() != 4 or t.shape[0] < 1 or tuple(t.shape[1:]) != (3,) + self._input_shape:
                    raise ValueError(f"expected a letterboxed [k,3,{self._input_shape[0]},{self._input_shape[1]}] array, got {tuple(t.shape)}")
                x[i].copy_(t[0], non_blocking=True)
        return x

    def run_device(self, imgs, max_det=1000):
        """``run`` without the download: -> (rows [n, max_det, 6] float32, counts [n] int32), both on the device, nothing
        synchronised — rows[i, :counts[i]] = (x1, y1, x2, y2, conf, cls) of entry i.  The letterboxed images go through the network
        in sub-batches of 16, each followed by ONE batched NMS call on the current stream."""
        eng = self._eng_net
        if not isinstance(imgs, (list, tuple)):
            imgs = [imgs]
        n = len(imgs)
        rows = torch.empty((n, max_det, 6), dtype=torch.float32, device=eng.device)
        counts = torch.zeros(n, dtype=torch.int32, device=eng.device)
        if n == 0:
            return rows, counts
        x = self._letterboxed(imgs)
        for b0 in range(0, n, 16):
            pred = eng.forward(x[b0:b0 + 16])
            eng.nms_batch_async(pred, self._conf_thresh, self._iou_thresh, max_det=max_det, out=rows[b0:b0 + 16], cnt=counts[b0:b0 + 16])
        return rows, counts

    def run(self, imgs):
        """imgs: list of image paths, of HWC uint8 arrays (RGB), or of already letterboxed float32 [1,3,H,W] arrays (what
        ``load_localizer_img`` returns in the reference) -> list (one per entry) of CPU tensors [n,6] (x1, y1, x2, y2, conf, cls).
        (The reference runs the session image by image, localizer_engine.py:52; here: batched network, ONE synchronisation.)"""
        rows, counts = self.run_device(imgs)
        if rows.shape[0] == 0:
            return []
        counts = counts.cpu().tolist()
        rows = rows[:, : max(counts + [0])].cpu()
        return [rows[i, :c].clone() for i, c in enumerate(counts)]
