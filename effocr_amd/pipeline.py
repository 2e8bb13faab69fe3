"""Host-side glue of the recognizer hot path: batching, id -> character mapping and the
encode -> normalise -> k-NN -> characters sequence, with the reference's exact conventions.

Restated reference logic (file:line into the reference tree):
  create_batches / iteration      infer_effocr_onnx_multi.py:143-163
  candidate chars / blacklist     infer_effocr.py:203-205,209-212 ; infer_effocr_onnx_multi.py:502-510
  kNN branch of EffOCR.infer      infer_effocr.py:310-319,337-338
  ONNX-driver recognizer phase    infer_effocr_onnx_multi.py:350-375
"""
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
        return indices_to_chars(indices, self.candidate_chars)

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
    dev = knn_func.index.device
    embs = [l2_normalize(torch.from_numpy(e[0][0]).to(dev)) for e in embeddings]
    indices = [knn_func(e, k=1)[1] for e in embs]
    index_list = [ix.squeeze(-1).tolist() for ix in indices]
    flat = [item for sub in index_list for item in sub]
    return [candidate_chars[i] for i in flat], flat
