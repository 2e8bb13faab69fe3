"""Generates tests/golden/ref_hostlogic.json, ref_hostlogic.npz, ref_run_effocr.json and ref_run_effocr.npz by IMPORTING THE REFERENCE
ITSELF (/root/reference, this container only) and recording inputs / outputs of its own host-logic functions.  Only the
recorded DATA is committed; no reference file travels.

What can and cannot be imported.  The arithmetic of the hot path lives in packages that are absent here (timm, faiss,
onnxruntime, torchvision, cv2, ...), but the reference's HOST LOGIC is plain Python / torch / numpy / PIL.  An import-only
stub finder stands in for the absent top-level packages so that ``import infer_effocr_onnx_multi`` / ``infer_effocr`` /
``onnx_engines.localizer_engine`` / ``utils.datasets_utils`` succeed; once the imports are done the stubs are ARMED: any call
or attribute access on a stub raises, so no stub value can reach a recorded output.  The single exception are explicit
RECORDING SINKS (``T.Pad`` for MedianPad, ``cv2.resize`` / ``cv2.copyMakeBorder`` for letterbox): they record the ARGUMENTS
the reference computed (the pad / border geometry) and return a poisoned object whose value is never recorded.

Recorded (reference file:line):
  create_batches              infer_effocr_onnx_multi.py:143-158   (None -> zero image, literal-64 padding, batch_size != 64 quirks)
  iteration                   infer_effocr_onnx_multi.py:161-163
  en_preprocess               infer_effocr_onnx_multi.py:70-90     (ONNX driver form: tensors rows, no scores)
  en_postprocess              infer_effocr_onnx_multi.py:92-131    (LARGE_NUMBER 1e9, anchor_margin cases, assertion cases)
  jp_preprocess               infer_effocr_onnx_multi.py:134-140
  EffOCR.en_preprocess / en_postprocess / jp_preprocess   infer_effocr.py:345-418 (torch driver form: scores + thresholds)
  EffLocalizer.xywh2xyxy / box_iou / letterbox geometry   onnx_engines/localizer_engine.py:107-169
  MedianPad pad geometry      utils/datasets_utils.py:67-88
  EffOCR.infer                infer_effocr.py:255-343   the torch driver's per-line function incl. the kNN branch (:310-319, k = 10),
                              unbound on a namespace, localizer call replaced by preset results, oracle-backed transform / encoder / k-NN
  run_effocr                  infer_effocr_onnx_multi.py:227-397   the WHOLE ONNX driver function, run over duck-typed engines whose
                              arithmetic is the oracle's (localizer = preset NMS rows, char_transform = oracle/crop_transform_ref,
                              recognizer = oracle/encoders_ref vit_tiny_test, knn_func = oracle/flat_ip.c): every line of the
                              reference's glue — label split, sorts, torch.round / int(round(x * W / 640)) scaling, numpy slicing,
                              TransformationThread failure -> None -> zero image, create_batches, normalize, k=1 flatten,
                              per-line slicing, en_postprocess — produces the expected strings.

Run:  python tests/golden/make_ref_golden.py        (needs /root/reference; the tests only read the fixtures)
"""
import hashlib
import importlib.abc
import importlib.machinery
import io
import json
import os
import sys
import types
import contextlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

ABSENT = {"torchvision", "pytorch_metric_learning", "faiss", "mmcv", "deepsparse", "timm", "albumentations", "kornia",
          "onnxruntime", "onnx", "cv2", "nltk", "symspellpy", "detectron2", "mmdet", "pycocotools", "Levenshtein", "editdistance",
          "fuzzywuzzy", "omegaconf", "wandb", "transformers_stub_never"}
STATE = {"armed": False, "sinks": {}, "log": []}


class StubUsed(RuntimeError):
    pass


class _Poison:
    """What a recording sink returns: any use raises."""
    def __getattr__(self, name):
        raise StubUsed(f"value produced by a stub was used (.{name})")

    def __call__(self, *a, **k):
        raise StubUsed("value produced by a stub was called")


class _Stub:
    def __init__(self, name):
        object.__setattr__(self, "_name", name)

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = f"{self._name}.{name}"
        if STATE["armed"] and full not in STATE["sinks"] and not any(s.startswith(full + ".") for s in STATE["sinks"]):
            raise StubUsed(f"stub attribute {full} accessed while recording")
        return _Stub(full)

    def __call__(self, *args, **kwargs):
        if self._name in STATE["sinks"]:
            return STATE["sinks"][self._name](*args, **kwargs)
        if STATE["armed"]:
            raise StubUsed(f"stub {self._name} called while recording")
        return _Stub(self._name + "()")

    def __mro_entries__(self, bases):          # ``class X(stub.Base)`` at import time
        return (object,)

    def __iter__(self):                        # module-level ``zip(IMAGENET_DEFAULT_MEAN, ...)`` at import time: empty
        if STATE["armed"]:
            raise StubUsed(f"stub {self._name} iterated while recording")
        return iter(())


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(_Stub(self.__name__), name)


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def import_reference():
    for top in list(ABSENT):
        try:
            __import__(top)
            ABSENT.discard(top)                 # really installed: never stub a real package
        except Exception:
            pass
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            import infer_effocr_onnx_multi as multi
            import infer_effocr as single
            from onnx_engines.localizer_engine import EffLocalizer
            import utils.datasets_utils as du
    finally:
        os.chdir(cwd)
    STATE["armed"] = True
    return multi, single, EffLocalizer, du


# ---------------------------------------------------------------------------------------------------------------------
def _rows(rng, n, lo=0.0, hi=640.0, h=40.0, integral=False):
    x0 = rng.uniform(lo, hi, n)
    w = rng.uniform(2.0, 60.0, n)
    y0 = rng.uniform(0.0, h / 2, n)
    y1 = y0 + rng.uniform(4.0, h, n)
    b = np.stack([x0, y0, x0 + w, y1], 1)
    if integral:
        b = np.round(b)
    return b.astype(np.float32)


def record_create_batches(multi):
    cases = []
    for n, none_at, bs in [(0, [], 64), (1, [], 64), (1, [0], 64), (63, [5], 64), (64, [], 64), (65, [64], 64), (69, [3, 68], 64),
                           (128, [], 64), (130, [0, 129], 64), (40, [7], 32), (170, [], 100)]:
        data = [None if i in none_at else torch.full((3, 224, 224), float(i + 1)) for i in range(n)]
        out = multi.create_batches(data) if bs == 64 else multi.create_batches(data, batch_size=bs)
        assert all(isinstance(b, np.ndarray) for b in out)
        cases.append({"n": n, "none_at": none_at, "batch_size": bs, "rule": "crop i = full((3,224,224), i+1); None at none_at",
                      "shapes": [list(b.shape) for b in out], "dtypes": [str(b.dtype) for b in out],
                      "row_first": [[float(v) for v in b[:, 0, 0, 0]] for b in out],
                      "row_is_constant": [bool((b == b[:, :1, :1, :1]).all()) for b in out]})
    return cases


def record_iteration(multi):
    class M:
        def run(self, x):
            return [x * 2]
    m = M()
    a = np.arange(6, dtype=np.float32)
    out = multi.iteration(m, a)
    return {"is_tuple": isinstance(out, tuple), "len": len(out), "same_object": out[0] is out[1], "inner_len": len(out[0]),
            "value": out[0][0].tolist()}


def record_multi_prepost(multi, rng):
    pre, post, jp = [], [], []
    for c in range(220):
        nc, nw = int(rng.integers(1, 14)), int(rng.integers(0, 5))
        vertical = bool(c % 7 == 0)
        cb = _rows(rng, nc, integral=(c % 3 == 0))
        wb = _rows(rng, nw, integral=(c % 3 == 0))
        if c % 5 == 0 and nc > 1:                                  # exact ties in the sort key and in the distances
            cb[1, 0] = cb[0, 0]
            cb[1, 2] = cb[0, 2]
        if c % 11 == 0 and nw:                                    # a word right of every character: repeats closest_idx
            wb[-1, 0] = 700.0
        chars_t, words_t = torch.from_numpy(cb), torch.from_numpy(wb)
        s, wei = multi.en_preprocess(chars_t, words_t, vertical=vertical) if vertical else multi.en_preprocess(chars_t, words_t)
        pre.append({"chars": cb.tolist(), "words": wb.tolist(), "vertical": vertical,
                    "sorted_chars": [[float(v) for v in x] for x in s], "word_end_idx": [int(i) for i in wei]})
        sj = multi.jp_preprocess(chars_t, vertical=vertical)
        jp.append({"chars": cb.tolist(), "vertical": vertical, "sorted_chars": [[float(v) for v in x] for x in sj]})
    alphabet = list("aenrwuosvcxzWUOSVCXZTHEqkb-.,'1")
    for c in range(320):
        n = int(rng.integers(0, 16))
        line = "".join(rng.choice(alphabet, n)) if n else ""
        if c % 9 == 0 and n:
            line = " " * int(rng.integers(1, 3)) + line[int(rng.integers(1, 3)):] if n > 3 else line       # rare: whitespace glyphs
            line = line[:n].ljust(n, "a")
        k = int(rng.integers(0, 4))
        wei = sorted(set(int(v) for v in rng.integers(0, max(n, 1), k))) if (n and c % 13) else ([] if c % 2 else [0])
        base = float(rng.uniform(10, 30))
        heights = [float(base * (1.0 if ch in "aenrwuosvcxz-.,'" else 1.6) * rng.uniform(0.9, 1.1)) for ch in line]
        bottoms = [float(40.0 + rng.uniform(-1, 1) - (base * 0.5 if (ch == "-" and rng.uniform() < 0.5) else 0.0)) for ch in line]
        if c % 17 == 0 and n:
            bottoms[0] = 0.0                                        # the "charbottoms[0] == 0" quirk
        margin = [None, 0.05, 0.15, 0.3][c % 4]
        mult = 4 if c % 6 else 2
        case = {"line": line, "word_end_idx": wei, "heights": heights, "bottoms": bottoms, "anchor_margin": margin, "anchor_multiplier": mult}
        try:
            kw = {} if margin is None else {"anchor_margin": margin}
            if mult != 4:
                kw["anchor_multiplier"] = mult
            out = multi.en_postprocess(line, wei, heights, bottoms, **kw)
            case["out"] = out
        except AssertionError:
            case["raises"] = "AssertionError"
        except IndexError:
            case["raises"] = "IndexError"
        post.append(case)
    return pre, post, jp


def record_single_prepost(single, rng):
    """EffOCR's methods (infer_effocr.py:345-418) called unbound on a namespace carrying the attributes __init__ sets (:225-241)."""
    pre, post, jp = [], [], []
    for c in range(160):
        ns = types.SimpleNamespace(vertical=bool(c % 5 == 0), score_thresh=[0.5, 0.3][c % 2], score_thresh_word=[0.5, 0.7][c % 3 == 0],
                                   LARGE_NUM=1_000_000, anchor_multiplier=4, anchor_margin=[None, 0.1, 0.2][c % 3], spell_check=False)
        nc, nw = int(rng.integers(1, 12)), int(rng.integers(0, 5))
        cb = np.concatenate([_rows(rng, nc).astype(np.float64), rng.uniform(0.2, 1.0, (nc, 1))], 1)
        wb = np.concatenate([_rows(rng, nw).astype(np.float64), rng.uniform(0.2, 1.0, (nw, 1))], 1)
        wrapped = c % 2 == 0
        result = [[cb, wb]] if wrapped else [cb, wb]              # both forms :347 accepts
        s, wei = single.EffOCR.en_preprocess(ns, result)
        pre.append({"vertical": ns.vertical, "score_thresh": ns.score_thresh, "score_thresh_word": ns.score_thresh_word, "wrapped": wrapped,
                    "chars": cb.tolist(), "words": wb.tolist(), "sorted_chars": [[float(v) for v in x] for x in s],
                    "word_end_idx": [int(i) for i in wei]})
        sj = single.EffOCR.jp_preprocess(ns, [[cb]])
        jp.append({"vertical": ns.vertical, "score_thresh": ns.score_thresh, "chars": cb.tolist(),
                   "sorted_chars": [[float(v) for v in x] for x in sj]})
        n = int(rng.integers(0, 14))
        alphabet = list("aenrwuosvcxzWUOSVCXZTHE-.")
        line = "".join(rng.choice(alphabet, n)) if n else ""
        wl = sorted(set(int(v) for v in rng.integers(0, max(n, 1), int(rng.integers(0, 4))))) if n else []
        base = float(rng.uniform(10, 30))
        heights = [float(base * (1.0 if ch in "aenrwuosvcxz-." else 1.6) * rng.uniform(0.9, 1.1)) for ch in line]
        bottoms = [float(40.0 + rng.uniform(-1, 1)) for ch in line]
        case = {"line": line, "word_end_idx": wl, "heights": heights, "bottoms": bottoms, "anchor_margin": ns.anchor_margin}
        try:
            case["out"] = single.EffOCR.en_postprocess(ns, line, wl, heights, bottoms)
        except AssertionError:
            case["raises"] = "AssertionError"
        post.append(case)
    return pre, post, jp


def record_localizer_static(EffLocalizer, rng, arrays):
    x = rng.uniform(0, 640, (64, 6)).astype(np.float32)
    arrays["xywh_in"] = x
    arrays["xywh_out_torch"] = EffLocalizer.xywh2xyxy(torch.from_numpy(x)).numpy()
    arrays["xywh_out_numpy"] = EffLocalizer.xywh2xyxy(x.copy())
    a = EffLocalizer.xywh2xyxy(torch.from_numpy(rng.uniform(0, 640, (40, 4)).astype(np.float32)))
    b = EffLocalizer.xywh2xyxy(torch.from_numpy(rng.uniform(0, 640, (30, 4)).astype(np.float32)))
    b[:5] = a[:5]                                                  # identical boxes: IoU = 1 / (1 + eps')
    arrays["iou_a"], arrays["iou_b"] = a.numpy(), b.numpy()
    arrays["iou_out"] = EffLocalizer.box_iou(a, b).numpy()
    # letterbox geometry through recording sinks
    cases = []
    rec = {}

    def fake_resize(im, new_unpad, interpolation=None):
        rec["resize"] = [int(new_unpad[0]), int(new_unpad[1])]
        return _Poison()

    def fake_border(im, top, bottom, left, right, btype, value=None):
        rec["border"] = [int(top), int(bottom), int(left), int(right)]
        rec["value"] = [int(v) for v in value]
        return _Poison()
    STATE["sinks"]["cv2.resize"] = fake_resize
    STATE["sinks"]["cv2.copyMakeBorder"] = fake_border
    STATE["sinks"]["cv2.INTER_LINEAR"] = None
    STATE["sinks"]["cv2.BORDER_CONSTANT"] = None
    shapes = [(256, 4096), (64, 640), (48, 400), (640, 640), (480, 640), (1000, 30), (33, 777), (639, 641), (1, 1), (320, 320), (700, 900)]
    for (h, w) in shapes:
        for kw in ({"auto": False}, {"auto": True}, {"auto": False, "scaleup": False}, {"auto": False, "scaleFill": True},
                   {"auto": False, "new_shape": (640, 480)}, {"auto": False, "new_shape": 512}):
            rec.clear()
            im = np.zeros((h, w, 3), np.uint8)
            _, ratio, (dw, dh) = EffLocalizer.letterbox(im, **kw)
            cases.append({"shape": [h, w], "kwargs": {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()},
                          "ratio": [float(ratio[0]), float(ratio[1])], "dw": float(dw), "dh": float(dh),
                          "resize": rec.get("resize"), "border": rec["border"], "value": rec["value"]})
    for k in ("cv2.resize", "cv2.copyMakeBorder", "cv2.INTER_LINEAR", "cv2.BORDER_CONSTANT"):
        del STATE["sinks"][k]
    return cases


def record_medianpad(du, rng):
    rec = {}

    def fake_pad(padding, fill=0, padding_mode="constant"):
        rec["padding"], rec["fill"] = [int(v) for v in padding], [int(v) for v in fill]
        return lambda image: _Poison()
    STATE["sinks"]["torchvision.transforms.Pad"] = fake_pad
    cases = []
    for (h, w) in [(256, 17), (17, 256), (31, 31), (1, 9), (9, 1), (224, 223)]:
        im = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        for override in [(255, 255, 255), None]:
            rec.clear()
            du.MedianPad(override=override)(im)
            case = {"shape": [h, w], "seed_rule": "see image_sha256", "override": list(override) if override else None,
                    "padding_left_top_right_bottom": rec["padding"], "fill": rec["fill"]}
            if override is None:
                case["image"] = im.tolist() if h * w <= 300 else None
            cases.append(case)
    del STATE["sinks"]["torchvision.transforms.Pad"]
    return [c for c in cases if not (c["override"] is None and c.get("image") is None)]


# ---------------------------------------------------------------------------------------------------------------------
# run_effocr over oracle-backed engines
CHARS = list("aenrwuosvcxzTHEQUICKBROWN-") + [chr(0x4E00 + i) for i in range(70)]
# REF_ARCH=vit_small_patch16_224 records the SAME driver cases over the real-size encoder (oracle A's ViT-S/16, fp32) into
# ref_run_effocr_vits.{json,npz}: the strings of the reference's run_effocr / EffOCR.infer with BASELINE configs[1]'s architecture
ARCH, SIZE, SEED_ENC = os.environ.get("REF_ARCH", "vit_tiny_test"), 224, 21
SUFFIX = {"vit_tiny_test": "", "vit_small_patch16_224": "_vits"}[ARCH]


def line_image(seed, H, W):
    """Blocky synthetic line: 8 x 8 px cells of 8 grey levels x 3 channels (crops then differ strongly from one another)."""
    rng = np.random.RandomState(seed)
    cells = rng.randint(0, 8, ((H + 7) // 8, (W + 7) // 8, 3)).astype(np.uint8) * 32
    return np.ascontiguousarray(np.kron(cells, np.ones((8, 8, 1), np.uint8))[:H, :W])


def driver_cases():
    """(lang, vertical, [(H, W, seed, rows[n,6])]) — rows are what NMS hands over: x0,y0,x1,y1 in the 640 letterbox space, conf, label."""
    rng = np.random.RandomState(77)
    cases = []

    def rows(n_char, n_word, vertical=False, span=640.0):
        out = []
        pos = np.sort(rng.uniform(4, span - 30, n_char))
        for p in pos:
            w = rng.uniform(6, 22)
            a, b = p, p + w
            t0, t1 = rng.uniform(2, 12), rng.uniform(24, 38)
            box = [t0, a, t1, b] if vertical else [a, t0, b, t1]
            out.append(box + [rng.uniform(0.5, 0.99), 0.0])
        for _ in range(n_word):
            i = rng.randint(0, max(n_char, 1))
            left = (pos[i] if n_char else 10.0) - rng.uniform(0.5, 3.0)
            box = [2.0, left, 38.0, left + 80] if vertical else [left, 2.0, left + 80, 38.0]
            out.append(box + [rng.uniform(0.5, 0.99), 1.0])
        r = np.asarray(out, np.float32).reshape(-1, 6)
        return r[rng.permutation(len(r))]                           # NMS order is by confidence, not by position
    en = [(64, 640, 101, rows(14, 3)), (256, 4096, 102, rows(40, 6)), (48, 400, 103, rows(9, 2)), (64, 640, 104, rows(0, 2)),
          (64, 640, 105, np.zeros((0, 6), np.float32))]
    # edge rows: zero-width after rounding (-> failed transform -> zero image), negative / beyond-the-image coordinates
    # (numpy slice semantics), x.5 roundings (half to even in torch.round AND in Python's round), a box right of every word
    e = rows(10, 2)
    e[0, :4] = [100.2, 3.0, 100.4, 30.0]
    e[1, :4] = [-6.0, 3.0, 9.5, 30.0]
    e[2, :4] = [630.5, 3.0, 655.0, 30.0]
    e[3, :4] = [200.5, 3.0, 212.5, 30.0]
    e[4, :4] = [201.5, 3.0, 213.5, 31.0]
    en.append((64, 1000, 106, e))
    cases.append(("en", False, en))
    jp = [(64, 640, 201, rows(12, 0)), (256, 4096, 202, rows(33, 0)), (64, 640, 203, np.zeros((0, 6), np.float32)), (48, 400, 204, rows(70, 0))]
    cases.append(("jp", False, jp))
    jv = [(640, 64, 301, rows(12, 0, vertical=True)), (400, 48, 302, rows(7, 0, vertical=True, span=640.0))]
    cases.append(("jp", True, jv))
    return cases


def make_world():
    """What the duck-typed engines share: oracle-backed transform / encoder / k-NN, the glyph index and its characters."""
    sys.path.insert(0, ROOT)
    from effocr_amd.weights import init_state_dict
    from oracle import knn_ref
    from oracle.crop_transform_ref import paired_transform
    from oracle.encoders_ref import encoder_forward
    w = {"enc_sd": init_state_dict(ARCH, seed=SEED_ENC, img_size=SIZE), "seen": {}, "tag": (0, "en")}
    rng = np.random.RandomState(5)
    d = rng.standard_normal((32, {"vit_tiny_test": 128, "vit_small_patch16_224": 384}[ARCH])).astype(np.float32)
    w["index"] = np.ascontiguousarray(d / np.linalg.norm(d, axis=1, keepdims=True))
    w["chars"] = ["x"] * 32

    def char_transform(crop):                                       # stands in for create_paired_transform (torchvision absent)
        if crop.shape[0] == 0 or crop.shape[1] == 0:
            raise ValueError("empty crop")
        w["seen"].setdefault(hashlib.sha256(crop.tobytes() + bytes(str(crop.shape), "ascii")).hexdigest(), (w["tag"], crop.copy()))
        return torch.from_numpy(np.asarray(paired_transform(crop, size=SIZE), dtype=np.float32))

    def knn_func(embedding, k):                                     # PML FaissKNN.__call__ convention: (distances, indices) tensors
        dd, ii = knn_ref.flat_ip_search(embedding.numpy(), w["index"], k)
        return torch.from_numpy(dd), torch.from_numpy(ii)
    w["char_transform"], w["knn_func"] = char_transform, knn_func
    w["encode"] = lambda x: encoder_forward(ARCH, w["enc_sd"], x)
    return w


def build_index(w):
    """Glyph index = the oracle's embeddings of EVERY distinct crop the reference cut in pass 1 (each query then finds itself with
    score 1 and a top-1 margin of ~5e-3 — this miniature encoder's embeddings of different crops are 0.97-0.995 alike) + 16 random rows.
    Rows of 'en' cases get Latin letters (cycling; candidate chars need not be unique), the others CJK."""
    from oracle.encoders_ref import l2_normalize
    items = sorted(w["seen"].items(), key=lambda kv: (kv[1][0][0], kv[0]))
    xs = torch.stack([w["char_transform"](c) for _, (_, c) in items])
    emb = l2_normalize(w["encode"](xs)).numpy()
    rng = np.random.RandomState(6)
    d = rng.standard_normal((16, emb.shape[1])).astype(np.float32)
    w["index"] = np.ascontiguousarray(np.concatenate([emb, d / np.linalg.norm(d, axis=1, keepdims=True)]).astype(np.float32))
    latin = "aenrwuosvcxzTHEQUICKBROWN-"
    w["chars"] = [latin[i % 26] if tag[1] == "en" else chr(0x4E00 + i) for i, (_, (tag, _)) in enumerate(items)] + [chr(0x3041 + i) for i in range(16)]


def record_run_effocr(multi, tmpdir, w):
    from PIL import Image
    char_transform, index = w["char_transform"], w["index"]

    class Rec:
        def run(self, batch):
            assert isinstance(batch, np.ndarray) and batch.shape[1:] == (3, SIZE, SIZE)
            return [w["encode"](torch.from_numpy(batch)).numpy()]

    cases = driver_cases()
    multi.knn_func, multi.candidate_chars = w["knn_func"], w["chars"]    # module globals in the reference (:372,375; set in __main__ :496-505)

    out_cases, arrays = [], {}
    for ci, (lang, vertical, lines) in enumerate(cases):
        w["tag"] = (ci, lang)
        paths, by_path = [], {}
        for li, (H, W, seed, r) in enumerate(lines):
            p = os.path.join(tmpdir, f"c{ci}_l{li}.png")
            Image.fromarray(line_image(seed, H, W)).save(p)
            paths.append(p)
            by_path[p] = torch.from_numpy(r.copy())
            arrays[f"rows_{ci}_{li}"] = r

        class Loc:
            _model_backend = "yolo"

            def run(self, ps):
                return [by_path[ps[0]]]
        for margin in ([None, 0.15] if lang == "en" else [None]):
            # en_postprocess's anchor_margin is not reachable through run_effocr's signature: the reference calls it with the default
            # (None, :390); the margin case binds it the only way the reference allows, through the function's default value
            saved = multi.en_postprocess.__defaults__
            if margin is not None:
                multi.en_postprocess.__defaults__ = (margin,) + tuple(saved[1:])
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    res, coco = multi.run_effocr(paths, Loc(), Rec(), char_transform, lang, num_streams=3, vertical=vertical)
            finally:
                multi.en_postprocess.__defaults__ = saved
            out_cases.append({"case": ci, "lang": lang, "vertical": vertical, "anchor_margin": margin,
                              "lines": [{"H": H, "W": W, "seed": seed, "rows": f"rows_{ci}_{li}",
                                         "sha256": hashlib.sha256(line_image(seed, H, W).tobytes()).hexdigest()}
                                        for li, (H, W, seed, r) in enumerate(lines)],
                              "outputs": [res[p] for p in paths], "coco": coco})
    return out_cases, arrays


def record_infer(single, tmpdir, w):
    """``EffOCR.infer`` (infer_effocr.py:255-343), the torch driver's per-line function, called UNBOUND on a namespace that carries
    what ``__init__`` sets (:217-241).  The localizer call (``inference_detector``, mmdet — absent) is replaced by preset mmdet-style
    results; ``char_transform`` / ``recongizer_encoder`` / ``recognizer.knn_func`` are oracle-backed.  k = 10 (the default, :112,241)."""
    from PIL import Image
    char_transform, index, knn_func = w["char_transform"], w["index"], w["knn_func"]
    rng = np.random.RandomState(91)
    presets = {}
    single.inference_detector = lambda localizer, im: presets[im]     # mmdet.apis.inference_detector (:262): the localizer stage
    out = []
    min_gap = 1.0
    for ci, (lang, vertical, H, W, seed, nc, nw, margin) in enumerate([
            ("en", False, 48, 400, 401, 11, 3, None), ("en", False, 64, 640, 402, 16, 4, 0.15), ("jp", False, 64, 640, 403, 13, 0, None),
            ("jp", True, 400, 48, 404, 9, 0, None), ("en", False, 64, 640, 405, 0, 2, None), ("jp", False, 64, 640, 406, 5, 0, None)]):
        w["tag"] = (100 + ci, lang)
        im = line_image(seed, H, W)
        p = os.path.join(tmpdir, f"infer_{ci}.png")
        Image.fromarray(im).save(p)
        span = H if vertical else W
        pos = np.sort(rng.uniform(2, span - 30, nc))
        cb = []
        for q in pos:
            a, b = q, q + rng.uniform(6, 24)
            t0, t1 = rng.uniform(2, 10), rng.uniform(20, 40)
            cb.append(([t0, a, t1, b] if vertical else [a, t0, b, t1]) + [rng.uniform(0.3, 0.99)])
        wbx = [[pos[rng.randint(0, nc)] - rng.uniform(0.2, 3), 1.0, 0.0, 44.0, rng.uniform(0.3, 0.99)] for _ in range(nw)] if nc else \
              [[5.0, 1.0, 50.0, 44.0, 0.9] for _ in range(nw)]
        for w_ in wbx:
            w_[2] = w_[0] + 70.0
        cb = np.asarray(cb, np.float32).reshape(-1, 5)
        cb = cb[rng.permutation(len(cb))]
        wb = np.asarray(wbx, np.float32).reshape(-1, 5)
        if ci == 5:
            cb[0, :4] = [0.5, 2.0, 11.5, 30.0]                        # .5 -> Python's round: half to even (0, 12).  A NEGATIVE x0 would slice
            #                                                           an empty crop, ValueError -> the reference calls exit(1) (:294-297)
            cb[1, :4] = [630.0, 2.0, 660.0, 30.0]                     # beyond the right edge
        presets[p] = [cb, wb] if lang == "en" else [[cb]]
        ns = types.SimpleNamespace(d2=False, localizer=None, lang=lang, vertical=vertical, double_clipped=True, char_transform=char_transform,
                                   N_classes=None, device="cpu", knn=10, candidate_chars=w["chars"], spell_check=False, LARGE_NUM=1_000_000,
                                   anchor_multiplier=4, anchor_margin=margin, score_thresh=0.5, score_thresh_word=0.5,
                                   recongizer_encoder=w["encode"],
                                   recognizer=types.SimpleNamespace(knn_func=knn_func))
        ns.en_preprocess = lambda r, ns=ns: single.EffOCR.en_preprocess(ns, r)
        ns.jp_preprocess = lambda r, ns=ns: single.EffOCR.jp_preprocess(ns, r)
        ns.en_postprocess = lambda *a, ns=ns: single.EffOCR.en_postprocess(ns, *a)
        with contextlib.redirect_stdout(io.StringIO()):
            output, output_nns, char_bboxes, word_bboxes = single.EffOCR.infer(ns, p)
        if output_nns is not None:                                    # rank stability of the recorded top-10 lists under fp32 re-ordering
            kept = [c for c in sorted(cb.tolist(), key=lambda x: x[1] if vertical else x[0]) if c[4] > 0.5]
            xs = []
            for bb in kept:
                x0, y0, x1, y1 = map(int, map(round, bb[:4]))
                xs.append(char_transform(im[y0:y1, 0:W] if vertical else im[0:H, x0:x1]))
            from oracle.encoders_ref import l2_normalize
            e = l2_normalize(w["encode"](torch.stack(xs))).numpy()
            sc = np.sort(e @ index.T, axis=1)[:, ::-1][:, :11]
            gaps = (sc[:, :-1] - sc[:, 1:]).min(1)
            min_gap = min(min_gap, float(gaps.min()))
        else:
            gaps = np.zeros(0)
        out.append({"lang": lang, "vertical": vertical, "H": H, "W": W, "seed": seed, "anchor_margin": margin,
                    "sha256": hashlib.sha256(im.tobytes()).hexdigest(), "chars": cb.tolist(), "words": wb.tolist(),
                    "output": output, "output_nns": output_nns, "rank_gap": [float(g) for g in gaps],
                    "char_bboxes": None if char_bboxes is None else [[float(v) for v in b] for b in char_bboxes],
                    "word_bboxes": None if word_bboxes is None else [[float(v) for v in b] for b in word_bboxes]})
    return out, min_gap


def main():
    import tempfile
    multi, single, EffLocalizer, du = import_reference()
    rng = np.random.default_rng(2024)
    arrays = {}
    pre, post, jp = record_multi_prepost(multi, rng)
    spre, spost, sjp = record_single_prepost(single, rng)
    fixture = {
        "generated_by": "tests/golden/make_ref_golden.py (imports /root/reference; stubs armed while recording)",
        "LARGE_NUMBER_multi": multi.LARGE_NUMBER,
        "distinct_lowercase": multi.create_distinct_lowercase(), "nondistinct_lowercase": multi.create_nondistinct_lowercase(),
        "COCO_JSON_SKELETON": multi.COCO_JSON_SKELETON,
        "create_batches": record_create_batches(multi),
        "iteration": record_iteration(multi),
        "multi_en_preprocess": pre, "multi_en_postprocess": post, "multi_jp_preprocess": jp,
        "single_en_preprocess": spre, "single_en_postprocess": spost, "single_jp_preprocess": sjp,
        "letterbox": record_localizer_static(EffLocalizer, rng, arrays),
        "medianpad": record_medianpad(du, rng),
    }
    if not SUFFIX:                                           # (the host-logic fixtures do not depend on the encoder)
        with open(os.path.join(HERE, "ref_hostlogic.json"), "w") as f:
            json.dump(fixture, f, ensure_ascii=False, separators=(",", ":"))
        np.savez_compressed(os.path.join(HERE, "ref_hostlogic.npz"), **arrays)
    with tempfile.TemporaryDirectory() as tmp:
        w = make_world()
        record_run_effocr(multi, tmp, w)       # pass 1: only to see which crops the reference cuts
        record_infer(single, tmp, w)
        build_index(w)
        cases, arr = record_run_effocr(multi, tmp, w)
        infer_cases, min_gap = record_infer(single, tmp, w)
        arr["index"] = w["index"]
    with open(os.path.join(HERE, f"ref_run_effocr{SUFFIX}.json"), "w") as f:
        json.dump({"chars": w["chars"], "arch": ARCH, "size": SIZE, "enc_seed": SEED_ENC, "cases": cases,
                   "infer": infer_cases, "infer_min_rank_gap": min_gap}, f, ensure_ascii=False, indent=0)
    np.savez_compressed(os.path.join(HERE, f"ref_run_effocr{SUFFIX}.npz"), **arr)
    n = sum(len(v) for v in fixture.values() if isinstance(v, list))
    print(f"recorded {n} host-logic cases + {len(cases)} run_effocr runs from the imported reference")
    for c in cases:
        print(c["lang"], c["vertical"], c["anchor_margin"], c["outputs"])
    for c in infer_cases:
        print("infer", c["lang"], c["vertical"], c["anchor_margin"], repr(c["output"]), (c["output_nns"] or [None])[0])
    print("infer min rank gap", min_gap)


if __name__ == "__main__":
    main()
