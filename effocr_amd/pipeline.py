"""Host-side glue of the recognizer hot path: batching, id -> character mapping and the
encode -> normalise -> k-NN -> characters sequence, with the reference's exact conventions.

Restated reference logic (file:line into the reference tree):
  create_batches / iteration      infer_effocr_onnx_multi.py:143-163
  candidate chars / blacklist     infer_effocr.py:203-205,209-212 ; infer_effocr_onnx_multi.py:502-510
  kNN branch of EffOCR.infer      infer_effocr.py:310-319,337-338
  ONNX-driver recognizer phase    infer_effocr_onnx_multi.py:350-375
"""
import os
import queue
import threading

import numpy as np
import torch

from .knn import FaissKNN, InferenceModel


def create_batches(data, batch_size=64):
    """infer_effocr_onnx_multi.py:143-158.  ``None`` crops become zero images; full chunks of
    ``batch_size``; the last chunk is zero-padded to the literal 64 of the reference (:157) —
    so, like there, ``batch_size`` is effectively fixed at 64."""
    batches, batch = [], []
    for i, d in enumerate(data):
        batch.append(d if d is not None else torch.zeros((3, 224, 224)))
        if (i + 1) % batch_size == 0:
            batches.append(torch.stack(batch))
            batch = []
    if len(batch) > 0:
        batches.append(torch.nn.functional.pad(torch.stack(batch), (0, 0, 0, 0, 0, 0, 0, 64 - len(batch))))
    return [b.detach().numpy() for b in batches]


def iteration(model, input):
    """infer_effocr_onnx_multi.py:161-163: returns ``(output, output)``; consumers read [0][0]."""
    output = model.run(input)
    return output, output


def read_candidate_chars(path):
    """``ref.txt`` -> list of glyph strings (whitespace split, infer_effocr.py:203-205)."""
    with open(path) as f:
        return f.read().split()


def write_candidate_chars(chars, path):
    """train_effocr_recognizer.py:61-62: one glyph per line."""
    with open(path, "w") as f:
        f.write("\n".join(chars))


def apply_blacklist(knn_func, candidate_chars, blacklist):
    """infer_effocr.py:209-212: iterate the CHARACTERS of ``blacklist``, drop their rows from the
    index (compaction) and filter the char list in the same order.  Raises KeyError for a
    blacklisted char that is not in the list, like the reference's dict lookup."""
    if blacklist is None:
        return candidate_chars
    candidate_chars_dict = {c: idx for idx, c in enumerate(candidate_chars)}
    blacklist_ids = np.array([candidate_chars_dict[blc] for blc in blacklist], dtype=np.int64)
    knn_func.index.remove_ids(blacklist_ids)
    return [c for c in candidate_chars if c not in blacklist]


def indices_to_chars(indices, candidate_chars):
    """infer_effocr.py:318-319,337-338 for an index tensor [B,k] (k >= 2 in the reference):
    -> (nearest_chars list[B] of list[k], output_nns list[B] of str, output str)."""
    if indices.dim() == 2 and indices.shape[1] == 1:
        # the reference's ``squeeze(-1)`` turns k=1 into a 1-D list and then fails (SURVEY a-1); here
        # k=1 simply yields one neighbour per crop
        index_list = [[i] for i in indices[:, 0].cpu().tolist()]
    else:
        index_list = indices.squeeze(-1).cpu().tolist()
    nearest_chars = [[candidate_chars[nn] for nn in nns] for nns in index_list]
    output_nns = ["".join(chars).strip() for chars in nearest_chars]
    output = "".join(x[0] for x in nearest_chars).strip()
    return nearest_chars, output_nns, output


def encoder_device(encoder):
    """Device of an encoder object of this package (AutoEncoder look-alike or HipEncoder); None if unknown."""
    for attr in ("_device", "device"):
        d = getattr(encoder, attr, None)
        if d is not None:
            return torch.device(d)
    eng = getattr(encoder, "_engine", None)
    return getattr(eng, "device", None)


def check_encoder_status(encoder):
    """Raise (EFFOCR_EOVERFLOW) if the encoder's last forward on the current stream produced a non-finite embedding — an f16
    operand overflow cannot hide behind plausible ids.  No-op for encoders that are not of this package."""
    eng = getattr(encoder, "engine", None) or getattr(encoder, "_eng_net", None) or encoder
    chk = getattr(eng, "check_status", None)
    if chk is not None:
        chk()


class Recognizer:
    """The kNN branch of ``EffOCR.infer`` (infer_effocr.py:310-319) as one object:
    crops -> encoder -> L2 normalise (fused) -> IP top-k -> characters, everything on one GPU."""

    def __init__(self, encoder, knn_func, candidate_chars, knn=10):
        self.recongizer_encoder = encoder          # sic — attribute name of infer_effocr.py:224
        # crops are moved to the ENCODER's device (PML's InferenceModel uses the current device; a literal cuda:0
        # broke every rank > 0 of the one-process-per-GPU layout)
        self.recognizer = InferenceModel(encoder, knn_func=knn_func, data_device=encoder_device(encoder))
        self.candidate_chars = candidate_chars
        self.knn = knn

    def neighbors(self, crops):
        """crops: [B,3,H,W] float32 tensor (or list of [3,H,W]) -> (distances, indices) on device."""
        if isinstance(crops, (list, tuple)):
            crops = torch.stack(list(crops))
        emb = self.recognizer.get_embeddings(crops)                 # encode + F.normalize
        return self.recognizer.knn_func(emb, k=self.knn)

    def __call__(self, crops):
        _, indices = self.neighbors(crops)
        out = indices_to_chars(indices, self.candidate_chars)      # (.cpu(): the call's synchronisation point)
        check_encoder_status(self.recongizer_encoder)
        return out

    def recognize_boxes(self, image, char_bboxes, char_transform=None, double_clipped=True, vertical=False):
        """infer_effocr.py:281-319 with the crop loop moved to the device: `image` is the HWC uint8 page /
        line image, `char_bboxes` the localizer's (x0,y0,x1,y1[,score]) rows.  One uint8 upload, then
        crop + pad + resize + normalise (effocr_crop_transform), encode, normalise, top-k — no per-crop
        PCIe traffic.  Returns (nearest_chars, output_nns, output) as `EffOCR.infer` builds them
        (:319, :337-338): list[B] of list[k] of str, list[B] of str, str.
        ``double_clipped`` (reference: hard-coded True, infer_effocr.py:226) widens every box to the full line height
        (x0,0,x1,H) — or to the full width (0,y0,W,y1) when ``vertical`` — after rounding, exactly as :286-291."""
        from .transforms import PairedTransform
        if char_transform is None:
            enc = self.recongizer_encoder
            size = getattr(enc, "img_size", None) or getattr(getattr(enc, "_engine", None), "img_size", 224)
            char_transform = PairedTransform(size=size)
        if double_clipped:
            H, W = int(image.shape[0]), int(image.shape[1])
            clipped = []
            for bb in char_bboxes:
                x0, y0, x1, y1 = (int(round(float(v))) for v in list(bb)[:4])
                clipped.append((0, y0, W, y1) if vertical else (x0, 0, x1, H))
            crops = char_transform.boxes(image, clipped, already_int=True)
        else:
            crops = char_transform.boxes(image, char_bboxes)
        if crops.shape[0] == 0:
            return [], [], ""                                       # "No content detected!" (infer_effocr.py:304-306)
        return self(crops)


class RecognizerEngineExecutorThread(threading.Thread):
    """infer_effocr_onnx_multi.py:207-223: pulls ``(i, batch)`` pairs off a queue until it is empty, runs ``iteration``
    on the SHARED engine and posts ``(i, output)``.  (``get_nowait`` instead of the reference's ``empty()`` + blocking
    ``get()``, which can hang when two threads race for the last item.)"""

    def __init__(self, model, input_queue: queue.Queue, output_queue: queue.Queue):
        super().__init__()
        self._model, self._input_queue, self._output_queue = model, input_queue, output_queue
        self.error = None

    def run(self):
        while True:
            try:
                i, batch = self._input_queue.get_nowait()
            except queue.Empty:
                return
            try:
                self._output_queue.put((i, iteration(self._model, batch)))
            except Exception as e:                                  # surfaced by run_recognizer_batches
                self.error = e
                return


def run_recognizer_batches(char_crops, recognizer_engine, knn_func, candidate_chars, normalize=None, num_streams=1):
    """Recognizer phase of ``run_effocr`` (infer_effocr_onnx_multi.py:347-375): batches of 64 -> engine (on
    ``num_streams`` executor threads sharing the one engine, results re-ordered by batch index, :350-369) -> normalise
    -> knn(k=1) -> flat list of characters (the padded tail included, exactly like the reference, whose consumers
    never read it)."""
    from .knn import l2_normalize
    batches = create_batches(char_crops)
    if num_streams <= 1:
        embeddings = [iteration(recognizer_engine, b) for b in batches]
    else:
        input_queue, output_queue = queue.Queue(), queue.Queue()
        for i, b in enumerate(batches):
            input_queue.put((i, b))
        threads = [RecognizerEngineExecutorThread(recognizer_engine, input_queue, output_queue) for _ in range(num_streams)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for t in threads:
            if t.error is not None:
                raise t.error
        embeddings = [None] * len(batches)
        while not output_queue.empty():
            i, result = output_queue.get()
            embeddings[i] = result
    check_encoder_status(recognizer_engine)                 # EffRecognizer.run already refuses non-finite embeddings; other engines: here
    dev = knn_func.index.device
    embs = [l2_normalize(torch.from_numpy(e[0][0]).to(dev)) for e in embeddings]
    indices = [knn_func(e, k=1)[1] for e in embs]
    index_list = [ix.squeeze(-1).tolist() for ix in indices]
    flat = [item for sub in index_list for item in sub]
    return [candidate_chars[i] for i in flat], flat


# ---------------------------------------------------------------------------------------------------------------------
# run_effocr: the whole ONNX driver (infer_effocr_onnx_multi.py:227-397) with every array-sized step on the device

LARGE_NUMBER = 1_000_000_000          # infer_effocr_onnx_multi.py:46 (the torch driver's self.LARGE_NUM is 1_000_000, infer_effocr.py:239)
COCO_JSON_SKELETON = {"info": {"": ""}, "licenses": [{"": ""}], "images": [], "annotations": [],
                      "categories": [{"id": 0, "name": "char"}]}          # utils/coco_utils.py:3-9


def word_end_indices(char_rights, word_lefts):
    """``en_preprocess`` of the ONNX driver (infer_effocr_onnx_multi.py:70-90) after the two sorts: for every word box (left to
    right) the index of the character whose RIGHT edge lies right of the word's LEFT edge and closest to it; the strict ``<``
    keeps the first minimum and ``closest_idx`` is NOT reset between words (a word without a candidate repeats the previous
    index) — both quirks kept."""
    rights = np.asarray(char_rights, dtype=np.float64)
    out, closest = [], 0
    for wl in word_lefts:
        cand = np.nonzero(rights > wl)[0]
        if cand.size:
            d = np.abs(wl - rights[cand])
            if d.min() < LARGE_NUMBER:                                       # prev_dist starts at LARGE_NUMBER (:80)
                closest = int(cand[int(np.argmin(d))])
        out.append(closest)
    return out


_PREFETCH = None
_AUX_STREAMS = {}     # (device, role) -> the ONE side stream of that role: the engines keep a workspace per stream they have seen, so
                      # run_effocr must not hand them a fresh stream per call (torch deals streams from a pool of 32: 32 workspaces)


def _aux_stream(dev, role):
    key = (str(dev), role)
    st = _AUX_STREAMS.get(key)
    if st is None:
        st = _AUX_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st



def _prefetcher():
    global _PREFETCH
    if _PREFETCH is None:
        from concurrent.futures import ThreadPoolExecutor
        _PREFETCH = ThreadPoolExecutor(max_workers=1, thread_name_prefix="effocr-upload")
    return _PREFETCH


def _upload_lines(imgs, dev, stream=None):
    """HWC uint8 images of ONE geometry (numpy, or uint8 tensors already on ``dev``) -> ([L,H,W,3] device tensor, event).  Host
    images go to the device by the runtime's own pageable copy, line by line (round 6, tools/host_register_probe.py: 48.9 GB/s
    against 55.7 GB/s from pinned memory — the pageable -> pinned memcpy of the round-3..5 staging chain alone cost more than that
    difference).  ``stream``: issue the copies on this side stream (run_effocr's prefetch of the NEXT chunk of lines, from a helper
    thread: the copy blocks its caller with the GIL released); the returned event marks the last copy — consumers on another
    stream wait for it."""
    if all(isinstance(im, torch.Tensor) and im.device == dev for im in imgs):
        return torch.stack(list(imgs)).contiguous(), None
    L, shape = len(imgs), tuple(imgs[0].shape)
    with torch.cuda.device(dev), torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream(dev)):
        out = torch.empty((L,) + shape, dtype=torch.uint8, device=dev)
        for j, im in enumerate(imgs):
            out[j].copy_(im if isinstance(im, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(im)))
        ev = None
        if stream is not None:
            ev = torch.cuda.Event()
            ev.record(stream)
    return out, ev


def _char_boxes_torch(rows, counts, max_det, H, W, axis, vertical):
    """The same box stage as ~25 ATen launches (rounds 3-5): the fall-back for max_det > 4096 and what tests compare the kernel with."""
    dev, L = rows.device, int(rows.shape[0])
    valid = torch.arange(max_det, device=dev)[None, :] < counts[:, None]
    is_char = valid & (rows[..., 5] == 0)
    key = torch.where(is_char, rows[..., axis], torch.full_like(rows[..., axis], float("inf")))
    order = torch.sort(key, dim=1, stable=True).indices                  # sorted(bboxes_char, key=x[0] | x[1]) (:72,134): stable
    boxes = torch.gather(rows[..., :4], 1, order[..., None].expand(-1, -1, 4))     # [L,max_det,4], the first n_chars rows are characters
    n_chars = is_char.sum(1)
    r = torch.round(boxes).double()                                      # torch.round(bbox) (:313)
    sel = torch.arange(max_det, device=dev)[None, :] < n_chars[:, None]
    line_idx = torch.arange(L, device=dev, dtype=torch.int64)[:, None].expand(-1, max_det)
    if vertical:                                                         # (:315-316)
        lo = torch.round(r[..., 1] * H / 640).to(torch.int64)
        hi = torch.round(r[..., 3] * H / 640).to(torch.int64)
        y0, y1 = _resolve_slice(lo, H), _resolve_slice(hi, H)
        x0, x1 = torch.zeros_like(y0), torch.full_like(y0, W)
    else:                                                                # (:317-318)
        lo = torch.round(r[..., 0] * W / 640).to(torch.int64)
        hi = torch.round(r[..., 2] * W / 640).to(torch.int64)
        x0, x1 = _resolve_slice(lo, W), _resolve_slice(hi, W)
        y0, y1 = torch.zeros_like(x0), torch.full_like(x0, H)
    boxes5 = torch.stack((x0, y0, x1, y1, line_idx), dim=-1)[sel].to(torch.int32)      # [total,5]; boolean indexing = a host sync
    return boxes, n_chars.to(torch.int32), boxes5


def _char_boxes(rows, counts, max_det, H, W, axis, vertical, defer_total=False):
    """Box stage of run_effocr on the device (csrc/boxes.hip, effocr_parse_char_boxes): NMS rows [L, max_det, 6] + counts [L] ->
    (boxes [L, max_det, 4] with the characters first, stably sorted along the reading axis; n_chars [L] int32; boxes5 [total, 5] int32 =
    the crop slice of every character, compact over the lines).  One host read (the total): the crop tensor is allocated from it.
    ``defer_total``: return (boxes, n_chars, the un-cut [L * max_det, 5] buffer, the device total) without reading anything."""
    from . import _lib
    if max_det > 4096:
        out = _char_boxes_torch(rows, counts, max_det, H, W, axis, vertical)
        return out + (None,) if defer_total else out
    L = int(rows.shape[0])
    dev = rows.device
    Lh = _lib.lib()
    rows = rows.contiguous()
    counts = counts.to(torch.int32).contiguous()
    boxes = torch.empty((L, max_det, 4), dtype=torch.float32, device=dev)
    n_chars = torch.empty(L, dtype=torch.int32, device=dev)
    b5 = torch.empty((L * max_det, 5), dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(Lh.effocr_parse_char_boxes(_lib.ptr(rows), _lib.ptr(counts), L, int(max_det), int(H), int(W), int(axis), 1 if vertical else 0,
                                              _lib.ptr(boxes), _lib.ptr(n_chars), _lib.ptr(b5), _lib.ptr(total), _lib.current_stream(dev)),
                   "effocr_parse_char_boxes", Lh)
    if defer_total:                                      # (run_effocr reads the total later: the next launches go out first)
        return boxes, n_chars, b5, total
    return boxes, n_chars, b5[: int(total.item())]


def _draw_localizer_boxes(out_dir, coco_images, images, per_line, vertical):
    """``localizer_output`` of the ONNX driver (infer_effocr_onnx_multi.py:292-305): every line image saved under its base name with the
    character crop regions outlined in red — torch.round(bbox), scaled by size / 640 with Python's round(), full line height (width when
    vertical), NOT clipped (PIL clips what it draws).  In-memory images have no name: "<position>.png".  Debug output, host side."""
    from PIL import Image, ImageDraw
    for li, src in enumerate(coco_images):
        im = images[li]
        if isinstance(im, torch.Tensor):
            im = im.cpu().numpy()
        img = Image.fromarray(np.ascontiguousarray(im)).convert("RGB")
        W, H = img.size
        draw = ImageDraw.Draw(img)
        for bb in per_line[li][1]:
            x0, y0, x1, y1 = (float(torch.round(v)) for v in bb)
            if vertical:
                rect = (0, int(round(y0 * H / 640)), W, int(round(y1 * H / 640)))
            else:
                rect = (int(round(x0 * W / 640)), 0, int(round(x1 * W / 640)), H)
            draw.rectangle(rect, outline="red")
        img.save(os.path.join(out_dir, os.path.basename(src) if isinstance(src, str) else f"{li}.png"))


def _load_rgb(p):
    if isinstance(p, torch.Tensor):
        if p.dim() != 3 or p.shape[2] != 3 or p.dtype != torch.uint8:
            raise ValueError("run_effocr takes image paths or HWC uint8 RGB arrays / tensors")
        return p
    if isinstance(p, str):
        from PIL import Image
        return np.array(Image.open(p).convert("RGB"))                       # :311
    a = np.asarray(p)
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
        raise ValueError("run_effocr takes image paths or HWC uint8 RGB arrays")
    return a


def _resolve_slice(v, size):
    """numpy slice bounds on the device: a negative bound counts from the end, everything is clipped to [0, size]."""
    v = torch.where(v < 0, v + size, v)
    return v.clamp(0, size)


def run_effocr(coco_images, localizer_engine, recognizer_engine, char_transform, lang, num_streams=4, vertical=False,
               localizer_output=None, conf_thres=0.5, *, knn_func, candidate_chars, anchor_margin=None, max_det=1000, lines_per_chunk=16, overlap_localizer=True):
    """``run_effocr`` of infer_effocr_onnx_multi.py:227-397: text-line images -> {image key: transcription}.

    Same stages, same arithmetic, but the arrays never leave the GPU between them:
      localizer   every line image is uploaded ONCE (uint8), letterboxed, run through the YOLOv5s network in sub-batches and
                  NMS'd on the device (``EffLocalizer.run_device``; the reference: one ORT call per image on ``num_streams`` threads);
      boxes       class 0 = characters, class 1 = words (:252-256,275); characters sorted along the reading direction with a
                  STABLE sort in NMS order (``sorted(..., key=x[0])`` :72); every box rounded, scaled by size/640 — the literal
                  640 of :315-318 — rounded again (both half-to-even, like ``torch.round`` and Python's ``round``) and widened to
                  the full line height (width when ``vertical``), then resolved like the numpy slice ``im[y0:y1, x0:x1]`` (:320);
      crops       ONE ``effocr_crop_transform_batch`` launch over all boxes of all lines from the uploaded images (the reference
                  re-reads every image with PIL and transforms crop by crop on threads); an empty slice gives the zero image
                  ``create_batches`` substitutes for a failed transform (:145-147,196-200);
      recognizer  encoder + fused L2 normalise + k-NN (k = 1) over all crops (:347-375; the reference's zero padding of the
                  last batch of 64 produces rows nobody reads and is skipped);
      post        per line: characters joined (:384-385) and, for ``lang == "en"``, ``en_postprocess`` with the heights /
                  bottoms of the 640-space boxes (:323-325,387-392).
    Two host reads per chunk of lines (the number of crops; ids + boxes), the second one on an event of its own chunk: the next chunk's
    localizer is enqueued before the host looks at this chunk's results, so the device does not wait for Python between chunks.

    Differences from the reference signature: ``knn_func`` / ``candidate_chars`` are module globals there (:372,375) and
    keyword arguments here; ``anchor_margin`` is exposed (the reference call leaves ``en_postprocess``'s default None);
    ``num_streams`` / ``conf_thres`` are accepted and unused (HIP streams are ordered by the device; ``conf_thres`` only feeds the
    detectron2 / mmdetection branches); ``localizer_output``: a directory that receives the debug drawings of :292-305.  Results are keyed by
    the path (or by position for in-memory arrays) in INPUT order — the reference's order is thread-completion order.
    Images of different sizes are processed in groups of one geometry; HWC uint8 tensors already on the engines' device are
    taken as they are (no upload).  The reference hands over ALL line images of a job at once (:227, one list); here a group is
    processed in chunks of ``lines_per_chunk`` lines (bounded activation memory: ~70 crops per line) and the upload of chunk i+1
    runs on a side stream, from a helper thread, under the kernels of chunk i (round 6).  ``overlap_localizer``: the localizer + NMS + box
    stage of chunk i+1 run on a stream of their own BESIDE the recognizer of chunk i (its convolutions fill the CUs the encoder's
    partially filled rounds and launch ramps leave idle: 72.4 -> 67.8 ms per 64 lines, same box); results do not depend on it."""
    import copy
    from .postprocess import LinePostprocessor
    if lang not in ("en", "jp"):
        raise ValueError("lang must be 'en' or 'jp'")
    if getattr(localizer_engine, "_model_backend", "yolo") != "yolo":
        raise NotImplementedError("only the yolo localizer backend exists")
    keys = [p if isinstance(p, str) else i for i, p in enumerate(coco_images)]
    inference_results, inference_coco = {}, copy.deepcopy(COCO_JSON_SKELETON)
    if not coco_images:
        return inference_results, inference_coco
    if not hasattr(recognizer_engine, "encode_device"):
        raise TypeError("recognizer_engine must be an effocr_amd EffRecognizer (device-resident crops are handed to encode_device)")
    dev = recognizer_engine.device
    images = [_load_rgb(p) for p in coco_images]
    groups = {}
    for i, im in enumerate(images):
        groups.setdefault(tuple(im.shape[:2]), []).append(i)
    post = LinePostprocessor(lang=lang, vertical=vertical, anchor_margin=anchor_margin)
    # en_preprocess is called WITHOUT the vertical flag (:277: always sorts by x0); jp_preprocess gets it (:288)
    axis = 1 if (vertical and lang == "jp") else 0
    per_line = {}                                                            # line index -> (ids, sorted char boxes, word boxes)
    lpc = max(1, int(lines_per_chunk)) if lines_per_chunk else (1 << 30)
    chunks = [(hw, mem[c0:c0 + lpc]) for hw, mem in groups.items() for c0 in range(0, len(mem), lpc)]
    side = _aux_stream(dev, "upload") if len(chunks) > 1 else None
    cur = torch.cuda.current_stream(dev)
    fstream = _aux_stream(dev, "front") if (len(chunks) > 1 and overlap_localizer) else None
    uploads = {}                                                             # chunk index -> future of (stack, event) / the pair itself

    def start_upload(ci, prefetch):
        if ci >= len(chunks) or ci in uploads:
            return
        imgs = [images[i] for i in chunks[ci][1]]
        uploads[ci] = _prefetcher().submit(_upload_lines, imgs, dev, side) if prefetch else _upload_lines(imgs, dev)

    def front(ci):
        """Upload (waited for), localizer + NMS, box stage: everything of chunk ``ci`` that does not need the host.  Enqueued BEHIND
        the recognizer kernels of chunk ci-1 and BEFORE the host reads that chunk's results, so the device never waits for Python."""
        (H, W), members = chunks[ci]
        up = uploads.pop(ci)
        stack, ev = up.result() if hasattr(up, "result") else up
        fs = fstream if (fstream is not None and ci > 0) else cur          # chunk 0 has nothing to run beside
        with torch.cuda.stream(fs):
            if ev is not None:
                fs.wait_event(ev)
                stack.record_stream(fs)
            start_upload(ci + 1, prefetch=True)                              # the NEXT chunk's lines travel under this chunk's kernels
            L = len(members)
            rows, counts = localizer_engine.run_device([stack[j] for j in range(L)], max_det=max_det)
            boxes, n_chars, b5, total = _char_boxes(rows, counts, max_det, H, W, axis, vertical, defer_total=True)   # two launches
            fev = None
            if fs is not cur:
                fev = torch.cuda.Event()
                fev.record(fs)
        return dict(H=H, W=W, members=members, stack=stack, rows=rows, counts=counts, boxes=boxes, n_chars=n_chars, b5=b5, total=total, fev=fev)

    def back(st):
        """Crops -> encoder -> k-NN of a chunk (one host read: the number of crops), results on their way to pinned host memory."""
        if st["fev"] is not None:                                            # the front ran on the side stream: join, and tell the allocator
            cur.wait_event(st["fev"])
            for t in (st["stack"], st["rows"], st["counts"], st["boxes"], st["n_chars"], st["b5"], st["total"]):
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)
        if isinstance(st["total"], torch.Tensor):
            boxes5 = st["b5"][: int(st["total"].item())]                    # sync 1: the crop tensor is allocated from it
        else:
            boxes5 = st["b5"]
        if boxes5.shape[0]:
            # 16-bit hand-off (SURVEY f-2): the crops are written in the encoder's operand type — same embeddings bit for bit
            crops = char_transform.boxes_batch(st["stack"], boxes5, dtype=getattr(recognizer_engine, "crop_dtype", torch.float32))
            emb = recognizer_engine.encode_device(crops, normalize=True)
            ids = knn_func(emb, k=1)[1][:, 0]
        else:
            ids = torch.empty(0, dtype=torch.int64, device=dev)
        want = [ids, st["boxes"], st["n_chars"]] + ([st["rows"], st["counts"]] if lang == "en" else [])
        host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in want]   # (torch's caching pinned allocator)
        for h, t in zip(host, want):
            h.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(cur)
        st["host"], st["done"] = host, ev
        st["stack"] = None                                                   # (the line images are not needed any more)

    def finish(st):
        """Sync 2 — on THIS chunk's event only (the next chunk's localizer is already running) — and the string stage's inputs."""
        st["done"].synchronize()
        check_encoder_status(recognizer_engine)
        ids_h, boxes_h, n_h = st["host"][0].tolist(), st["host"][1], st["host"][2].tolist()
        if lang == "en":                                                     # word boxes = valid rows of class 1, in NMS order (:254-256)
            rows_h, counts_h = st["host"][3], st["host"][4].to(torch.int64)
            word_h = (torch.arange(max_det)[None, :] < counts_h[:, None]) & (rows_h[..., 5] == 1)
        off = 0
        for j, li in enumerate(st["members"]):
            n = n_h[j]
            wb = rows_h[j][word_h[j]][:, :4].clone() if lang == "en" else None
            per_line[li] = (ids_h[off:off + n], boxes_h[j, :n].clone(), wb)
            off += n

    start_upload(0, prefetch=False)
    st = front(0)
    for ci in range(len(chunks)):
        back(st)
        nxt = front(ci + 1) if ci + 1 < len(chunks) else None
        finish(st)
        st = nxt
    if localizer_output:
        _draw_localizer_boxes(localizer_output, coco_images, images, per_line, vertical)
    for li, key_ in enumerate(keys):
        ids_l, cb, wb = per_line[li]
        out = "".join(candidate_chars[i][0] for i in ids_l).strip()           # "".join(x[0] for x in textline).strip() (:385), k = 1
        if lang == "en":
            if len(ids_l):
                wl = sorted(float(v) for v in wb[:, 0])                      # sorted(bboxes_word, key=x[0]) lefts (:73,78)
                wei = word_end_indices(cb[:, 2].tolist(), wl)
            else:
                wei = []                                                     # (:283-285)
            heights = [float(b[3] - b[1]) for b in cb]                       # (:323-325)
            bottoms = [float(b[3]) for b in cb]
            out = post.en_postprocess(out, wei, heights, bottoms)
        inference_results[key_] = out
    return inference_results, inference_coco
