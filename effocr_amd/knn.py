"""k-NN engine: the ``faiss.IndexFlatIP`` + ``pytorch_metric_learning`` ``FaissKNN`` /
``InferenceModel`` call conventions of the reference, backed by the gfx950 inner-product top-k
kernel (effocr_amd/csrc/knn.hip) with the index resident in HBM.

Reference call sites kept working unchanged:
  infer_effocr.py:184-188   FaissKNN(index_init_fn=faiss.IndexFlatIP, reset_before=False, reset_after=False)
                            InferenceModel(recognizer_encoder, knn_func=knn_func)
  infer_effocr.py:201,207   recognizer.train_knn(render_dataset) / recognizer.load_knn_func(ref.index)
  infer_effocr.py:211       recognizer.knn_func.index.remove_ids(blacklist_ids)
  infer_effocr.py:317       _, indices = self.recognizer.knn_func(emb, k=self.knn)
  infer_effocr_onnx_multi.py:496-500,509,372   knn_func.load(...); knn_func.index.remove_ids(...); knn_func(emb, k=1)
  train_effocr_recognizer.py:27-29,35,49-52    load_knn_func / get_nearest_neighbors / train_knn / save_knn_func

Semantics: scores = fp32 inner products, k best per query in descending order,
int64 row ids; k > ntotal pads with id -1 / score -FLT_MAX; ``remove_ids`` deletes rows and
compacts.  Defined here because faiss leaves them open: the dot product is the ascending-k fmaf
chain and equal scores rank by ascending id.
"""
import struct
import threading

import numpy as np
import torch

from . import _lib

NEG_SCORE = -3.4028234663852886e+38


class IndexFlatIP:
    """HBM-resident exact inner-product index (faiss.IndexFlatIP role)."""

    metric_type = 0          # faiss.METRIC_INNER_PRODUCT
    is_trained = True

    SCREEN_WS_BYTES = 1 << 30   # bound on the pooled screen's block-maxima workspace per call (larger query batches are sliced)
    SCREEN_MIN_ROWS = 65536   # from this size on `search` uses the screened entry point (bit-identical results, ~4x faster at 1M rows)

    def __init__(self, d, device=None, screen="auto"):
        self.d = int(d)
        self.device = _lib.require_gpu(device)
        self._L = _lib.lib()
        self._xb = torch.empty((0, self.d), dtype=torch.float32, device=self.device)
        self._ws = {}                            # scratch per HIP stream: calls on different streams never share partial lists
        self.screen = screen                     # "auto" | True | False
        self.screen_overflows = 0                # screened searches that had to re-run exactly (candidate cap exceeded)
        self._flag_pending = None                # (pinned host copy of the device overflow flag, event) of the last screened call
        self._xb16 = None                        # bf16 copy of the rows + max row norm, built lazily for the screening pass
        self._xblk = None                        # fragment-blocked bf16 copy (effocr_convert_bf16_blocked): the Q-stationary screening pass
        self._xnorm_max = None
        self._copy_lock = threading.Lock()       # the lazy copies are built once, by whichever caller thread gets here first ...
        self._copy_ev = {}                       # ... and every OTHER stream waits for the builder's event before reading them

    @property
    def ntotal(self):
        return int(self._xb.shape[0])

    def _as_dev(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        if not isinstance(x, torch.Tensor):
            raise TypeError("expected a numpy array or torch tensor")
        if x.dim() != 2 or x.shape[1] != self.d:
            raise ValueError(f"expected shape [n,{self.d}], got {tuple(x.shape)}")
        return x.to(self.device, torch.float32).contiguous()

    def add(self, x):
        x = self._as_dev(x)
        self._xb = x.clone() if self.ntotal == 0 else torch.cat([self._xb, x], dim=0)
        self._drop_copies()

    def reset(self):
        self._xb = torch.empty((0, self.d), dtype=torch.float32, device=self.device)
        self._drop_copies()

    def _drop_copies(self):
        self._xb16 = self._xblk = self._xnorm_max = None
        self._copy_ev = {}

    SCREEN_MAX_OVERFLOWS = 3   # after this many overflowing calls screening is switched off for this index

    def _workspace(self, need):
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def _poll_overflow(self):
        """Non-blocking look at the previous screened call's device flag: an index full of near-duplicate rows (one
        glyph in many fonts) can exceed the 512-candidate cap, in which case that call ALSO ran the exact pass; after
        SCREEN_MAX_OVERFLOWS such calls the index stops screening (results are identical either way)."""
        pending = self._flag_pending             # (one read: another caller thread may clear the attribute between a test and an unpack)
        if pending is None:
            return
        host, ev = pending
        if not ev.query():
            return
        if self._flag_pending is pending:
            self._flag_pending = None
        if int(host.item()) != 0:
            self.screen_overflows += 1
            if self.screen == "auto" and self.screen_overflows >= self.SCREEN_MAX_OVERFLOWS:
                self.screen = False

    def _use_screen(self, k, nq=None):
        if self.screen is False or self.d % 64 != 0 or k > 32 or self.ntotal < max(k, 1):
            return False
        if self.screen is True:
            return True
        # Measured (tools/knn_sweep.py, 1M rows): up to 64 queries at d <= 384 (32 above) ONE launch of the exact streaming kernel beats
        # the screened search (0.72 vs 0.78 ms at 64 queries); beyond that screening wins (128 queries: 0.78 vs 1.42 ms; 1024: 3.2 vs 8.6 ms).
        # Small indexes: the bf16 pass + re-rank pays only for large batches (10 k rows, 1024 queries: 0.135 vs 0.164 ms).
        if nq is None:
            return self.ntotal >= self.SCREEN_MIN_ROWS
        # round 4: for 17..128 queries the screening pass streams the bf16 index once per 64 queries (dims that are multiples of 192):
        # 1M x 384, 64 queries: see profiles/README.md — from 17 queries on it beats the exact fp32 stream, which is MFMA-bound there
        if self.ntotal >= self.SCREEN_MIN_ROWS and self.d % 192 == 0 and self.d <= 768 and 16 < nq <= 128:
            return True
        stream_cap = 64 if self.d <= 384 else 32
        if self._qs_ok(k):
            # round 5, the Q-stationary pooled screen (tools/knn_c2_time.py, same box): 10 k x 384 at 1024 queries 88 us (k = 10) / 51 us
            # (k = 1) against 130 / 115 us on the round-4 chain and 160 / 97 us exact; 1M x 384 at 256 / 1024 queries 0.36 / 0.87 ms
            # against 0.99 / 3.2 ms; 1M x 768 at 1024 queries 1.81 against 3.96 ms.  Calls of <= 256 queries against a small index stay
            # exact (69 us: the chain's four dependent launches cost more than the products there).
            # (end of round 5, 4-row block maxima for small problems: 10 k x 384 at 64 / 128 / 256 / 1024 queries, k = 10: 51 / 53 / 54 / 71 us
            # against 69 / 69 / 71 / 159 us exact — the pooled chain wins from 33 queries on; k = 1, whose exact lists hold one entry:
            # 40 us exact up to 256 queries, 54 vs 97 us at 1024)
            return (self.ntotal >= self.SCREEN_MIN_ROWS and nq > stream_cap) or \
                   (self.ntotal >= 8192 and (nq >= 512 or (k > 1 and nq > 32)))
        # (k = 1 on a small index: the exact kernel keeps one-entry lists — 98 vs 111 us at 10 k rows x 1024 queries, tools/knn_c2_sweep.py)
        return (self.ntotal >= self.SCREEN_MIN_ROWS and nq > stream_cap) or (self.ntotal >= 8192 and nq >= 512 and k > 1)

    QS_DIMS = (128, 384, 768)  # embed dims of the Q-stationary screening kernel (ViT-tiny test width, ViT-S, ViT-B)
    use_qs = True              # A/B: False = screening passes without the blocked copy (round-4 kernels)

    def _qs_ok(self, k):
        return self.use_qs and self.d in self.QS_DIMS and k <= 16 and self.ntotal >= 1024

    def _screen_copy(self, rowmajor=True, blocked=False):
        """The bf16 copies of the rows a screening pass reads, built lazily per index change: row-major (the 128-query tile kernel,
        the streaming screen) and / or fragment-blocked (the Q-stationary kernel) — plus an upper bound of every row norm."""
        # One index is shared by N caller threads on their own HIP streams (the reference's driver: infer_effocr_onnx_multi.py:207-223).
        # The copies are built under a lock on the first caller's stream; an event recorded behind each build makes every other
        # stream wait for it — without it a second stream could screen against a half-written copy and return wrong ids.
        with self._copy_lock, torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            if rowmajor and self._xb16 is None:
                xb16 = torch.empty((self.ntotal, self.d), dtype=torch.bfloat16, device=self.device)
                _lib.check(self._L.effocr_convert_bf16(_lib.ptr(self._xb), self.ntotal * self.d, _lib.ptr(xb16),
                                                       _lib.current_stream(self.device)), "effocr_convert_bf16", self._L)
                ev = torch.cuda.Event()
                ev.record(cur)
                self._copy_ev["rowmajor"] = (ev, cur.cuda_stream)
                self._xb16 = xb16
            if blocked and self._xblk is None:
                xblk = torch.empty(int(self._L.effocr_bf16_blocked_bytes(self.ntotal, self.d)), dtype=torch.uint8, device=self.device)
                _lib.check(self._L.effocr_convert_bf16_blocked(_lib.ptr(self._xb), self.ntotal, self.d, _lib.ptr(xblk),
                                                               _lib.current_stream(self.device)), "effocr_convert_bf16_blocked", self._L)
                ev = torch.cuda.Event()
                ev.record(cur)
                self._copy_ev["blocked"] = (ev, cur.cuda_stream)
                self._xblk = xblk
            for kind, want in (("rowmajor", rowmajor), ("blocked", blocked)):
                rec = self._copy_ev.get(kind)
                if want and rec is not None and rec[1] != cur.cuda_stream:
                    cur.wait_event(rec[0])
            if self._xnorm_max is None:
                # an upper bound of every row norm (fp32 rounding slack included); one host read per index change
                self._xnorm_max = float(torch.linalg.vector_norm(self._xb, dim=1).max().item()) * (1.0 + 1e-5)
        return self._xb16

    def reconstruct_n(self, i0=0, n=None):
        n = self.ntotal - i0 if n is None else n
        return self._xb[i0:i0 + n].cpu().numpy()

    def remove_ids(self, ids):
        """Delete the given rows and compact (later rows shift down); returns the number removed."""
        ids = np.unique(np.asarray(ids, dtype=np.int64).reshape(-1))
        ids = ids[(ids >= 0) & (ids < self.ntotal)]
        if ids.size == 0:
            return 0
        keep = np.ones(self.ntotal, dtype=bool)
        keep[ids] = False
        rows = torch.from_numpy(np.nonzero(keep)[0].astype(np.int64)).to(self.device)
        dst = torch.empty((rows.numel(), self.d), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.effocr_gather_rows(_lib.ptr(self._xb), _lib.ptr(rows), rows.numel(), self.d,
                                                  _lib.ptr(dst), _lib.current_stream(self.device)), "effocr_gather_rows", self._L)
        self._xb = dst
        self._drop_copies()
        return int(ids.size)

    def search_device(self, q, k):
        """q: [n,d] float32 CUDA tensor -> (D [n,k] float32, I [n,k] int64) CUDA tensors, async."""
        k = int(k)
        if k <= 0:
            raise ValueError("k must be positive")
        q = self._as_dev(q)
        n = q.shape[0]
        D = torch.empty((n, k), dtype=torch.float32, device=self.device)
        I = torch.empty((n, k), dtype=torch.int64, device=self.device)
        if n == 0:
            return D, I
        self._poll_overflow()
        # The pooled screen's workspace holds one fp32 maximum per (16 index rows, query): ntotal / 16 * n * 4 bytes.  Very large query
        # batches against a large index go through it in slices of at most SCREEN_WS_BYTES of that (1M rows: 4 096 queries per slice).
        if self._use_screen(k, n) and self._qs_ok(k) and self.ntotal * n // 4 > self.SCREEN_WS_BYTES:
            per = max(512, (self.SCREEN_WS_BYTES * 4 // self.ntotal) // 256 * 256)
            if per < n:
                for lo in range(0, n, per):
                    d_, i_ = self.search_device(q[lo:lo + per], k)
                    D[lo:lo + per], I[lo:lo + per] = d_, i_
                return D, I
        if self._use_screen(k, n):
            qs = self._qs_ok(k)
            # the streaming screen (17..128 queries against a large index) reads the row-major copy; everything else the blocked one
            stream16 = self.ntotal >= self.SCREEN_MIN_ROWS and self.d % 192 == 0 and self.d <= 768 and 16 < n <= 128
            self._screen_copy(rowmajor=stream16 or not qs, blocked=qs and not stream16)
            need = int(self._L.effocr_knn_screen_workspace_bytes(n, self.ntotal, self.d, k))
            with torch.cuda.device(self.device):
                ws = self._workspace(need)
                _lib.check(self._L.effocr_knn_ip_topk_screened2(_lib.ptr(q), n, _lib.ptr(self._xb),
                                                                _lib.ptr(self._xb16 if (stream16 or not qs) else None),
                                                                _lib.ptr(self._xblk if (qs and not stream16) else None), self.ntotal, self.d, k,
                                                                self._xnorm_max, _lib.ptr(D), _lib.ptr(I), _lib.ptr(ws),
                                                                ws.numel(), _lib.current_stream(self.device)), "effocr_knn_ip_topk_screened", self._L)
                if self._flag_pending is None:
                    off = int(self._L.effocr_knn_screen_flag_offset(n, self.ntotal, self.d, k))
                    host = torch.empty(1, dtype=torch.int32).pin_memory()
                    host.copy_(ws[off:off + 4].view(torch.int32), non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
                    self._flag_pending = (host, ev)
            return D, I
        need = int(self._L.effocr_knn_workspace_bytes(n, self.ntotal, self.d, k))
        with torch.cuda.device(self.device):
            ws = self._workspace(need)
            _lib.check(self._L.effocr_knn_ip_topk(_lib.ptr(q), n, _lib.ptr(self._xb), self.ntotal, self.d, k,
                                                  _lib.ptr(D), _lib.ptr(I), _lib.ptr(ws), ws.numel(),
                                                  _lib.current_stream(self.device)), "effocr_knn_ip_topk", self._L)
        return D, I

    def search(self, x, k):
        """faiss signature: numpy in, numpy (D, I) out."""
        D, I = self.search_device(x, k)
        return D.cpu().numpy(), I.cpu().numpy()


# --------------------------------------------------------------------------- ref.index file format
# faiss ``write_index`` layout for IndexFlatIP, restated from faiss's index_write.cpp (faiss is not
# installable here, so this is UNVERIFIED against a file written by faiss itself):
#   fourcc "IxFI" u32 | d i32 | ntotal i64 | dummy i64 (1<<20) | dummy i64 (1<<20) | is_trained u8 |
#   metric_type i32 (0 = inner product) | n_floats u64 | ntotal*d float32 row-major
_FOURCC_IP = b"IxFI"


def write_index(index, path):
    xb = index._xb.cpu().numpy().astype("<f4", copy=False)
    with open(path, "wb") as f:
        f.write(_FOURCC_IP)
        f.write(struct.pack("<iqqqBi", index.d, index.ntotal, 1 << 20, 1 << 20, 1, 0))
        f.write(struct.pack("<Q", xb.size))
        f.write(xb.tobytes(order="C"))


_OTHER_FOURCC = {b"IxF2": "IndexFlatL2", b"IxFl": "legacy IndexFlat", b"IxF1": "IndexFlat1D", b"IwFl": "IndexIVFFlat",
                 b"IxPq": "IndexPQ", b"IHNf": "IndexHNSWFlat", b"IxMp": "IndexIDMap", b"IxM2": "IndexIDMap2"}


def read_index(path, device=None):
    with open(path, "rb") as f:                  # (device None / "cuda" = the current device: resolved by IndexFlatIP)
        buf = f.read()
    if buf[:4] != _FOURCC_IP:
        kind = _OTHER_FOURCC.get(bytes(buf[:4]))
        raise ValueError(f"{path}: not a faiss IndexFlatIP file (fourcc {bytes(buf[:4])!r}" + (f" = {kind}" if kind else "") +
                         "); the reference writes faiss.IndexFlatIP, 'IxFI' (train_effocr_recognizer.py:27,52)")
    hdr = struct.calcsize("<iqqqBi")
    if len(buf) < 4 + hdr + 8:
        raise ValueError(f"{path}: truncated IndexFlatIP header ({len(buf)} bytes)")
    d, ntotal, _, _, _, metric = struct.unpack_from("<iqqqBi", buf, 4)
    off = 4 + hdr
    (nfl,) = struct.unpack_from("<Q", buf, off)
    off += 8
    if d <= 0 or ntotal < 0 or metric != 0 or nfl != ntotal * d or len(buf) < off + 4 * nfl:
        raise ValueError(f"{path}: inconsistent IndexFlatIP header (d={d}, ntotal={ntotal}, metric={metric}, n={nfl}, "
                         f"{len(buf) - off} payload bytes)")
    idx = IndexFlatIP(d, device=device)
    if ntotal:
        idx.add(np.frombuffer(buf, dtype="<f4", count=nfl, offset=off).reshape(ntotal, d).copy())
    return idx


# --------------------------------------------------------------------------- PML look-alikes
class FaissKNN:
    """pytorch_metric_learning.utils.inference.FaissKNN call convention on the HIP index."""

    def __init__(self, reset_before=True, reset_after=True, index_init_fn=None, gpus=None, device=None):
        # device None = the current HIP device (what PML / faiss-gpu do); one process per GPU sets it with
        # torch.cuda.set_device(local_rank)
        if device is None and torch.cuda.is_available():
            device = torch.device("cuda", torch.cuda.current_device())
        self.reset_before = reset_before
        self.reset_after = reset_after
        # the reference passes faiss.IndexFlatIP; any callable taking d works, default = ours
        self.index_init_fn = IndexFlatIP if index_init_fn is None else index_init_fn
        self.gpus = gpus
        self.device = device
        self.index = None

    def _new_index(self, d):
        try:
            return self.index_init_fn(d, device=self.device)
        except TypeError:
            return self.index_init_fn(d)

    def __call__(self, query, k, reference=None, ref_includes_query=False):
        if ref_includes_query:
            k = k + 1
        device = query.device if isinstance(query, torch.Tensor) else torch.device("cpu")
        d = query.shape[1]
        if self.reset_before:
            self.index = self._new_index(d)
        if self.index is None:
            raise ValueError("self.index is None. It needs to be initialized before being used.")
        if reference is not None:
            self.index.add(reference)
        if isinstance(query, torch.Tensor):
            query = query.detach()
        distances, indices = self.index.search_device(query, k)
        distances, indices = distances.to(device), indices.to(device)
        if self.reset_after:
            self.reset()
        if ref_includes_query:
            distances, indices = distances[:, 1:], indices[:, 1:]
        return distances, indices

    def train(self, embeddings):
        self.index = self._new_index(embeddings.shape[1])
        self.add(embeddings)

    def add(self, embeddings):
        if isinstance(embeddings, torch.Tensor):
            embeddings = embeddings.detach()
        self.index.add(embeddings)

    def save(self, filename):
        write_index(self.index, filename)

    def load(self, filename):
        self.index = read_index(filename, device=self.device)

    def reset(self):
        if self.index is not None:
            self.index.reset()


class InferenceModel:
    """pytorch_metric_learning.utils.inference.InferenceModel subset used by the reference
    (infer_effocr.py:188,201,207,317; train_effocr_recognizer.py:27-35,49-52)."""

    def __init__(self, trunk, embedder=None, match_finder=None, normalize_embeddings=True, knn_func=None,
                 data_device=None, dtype=None):
        self.trunk = trunk
        self.embedder = embedder
        self.match_finder = match_finder
        self.normalize_embeddings = normalize_embeddings
        self.knn_func = FaissKNN(reset_before=False, reset_after=False) if knn_func is None else knn_func
        if data_device is None:                  # PML: the current device; here preferably the trunk's own device
            data_device = getattr(trunk, "_device", None) or getattr(trunk, "device", None)
            if data_device is None or isinstance(data_device, str) and data_device == "cuda":
                data_device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cuda"
        self.data_device = torch.device(data_device)
        self.dtype = dtype

    def get_embeddings(self, x):
        if isinstance(x, torch.Tensor):
            x = x.to(self.data_device, torch.float32)
        if hasattr(self.trunk, "eval"):
            self.trunk.eval()
        with torch.no_grad():
            eng = getattr(self.trunk, "engine", None)
            if eng is not None and self.embedder is None:
                return eng.forward(x, normalize=bool(self.normalize_embeddings))   # fused F.normalize
            emb = self.trunk(x)
            if self.embedder is not None:
                emb = self.embedder(emb)
        if self.normalize_embeddings:
            emb = l2_normalize(emb)
        return emb

    def _embed_all(self, inputs, batch_size):
        if isinstance(inputs, (list, tuple)):
            inputs = torch.stack(list(inputs))
        if isinstance(inputs, torch.Tensor):
            chunks = [self.get_embeddings(inputs[i:i + batch_size]) for i in range(0, len(inputs), batch_size)]
        else:   # a torch Dataset yielding (image, label) like FontImageFolder
            chunks, batch = [], []
            for i in range(len(inputs)):
                item = inputs[i]
                batch.append(item[0] if isinstance(item, (tuple, list)) else item)
                if len(batch) == batch_size or i == len(inputs) - 1:
                    chunks.append(self.get_embeddings(torch.stack(batch)))
                    batch = []
        return torch.cat(chunks, dim=0)

    def train_knn(self, inputs, batch_size=64):
        self.knn_func.train(self._embed_all(inputs, batch_size))

    def add_to_knn(self, inputs, batch_size=64):
        self.knn_func.add(self._embed_all(inputs, batch_size))

    def get_nearest_neighbors(self, query, k):
        return self.knn_func(self.get_embeddings(query), k)

    def save_knn_func(self, filename):
        self.knn_func.save(filename)

    def load_knn_func(self, filename):
        self.knn_func.load(filename)


def l2_normalize(x):
    """torch.nn.functional.normalize(x, p=2, dim=1) on device (infer_effocr.py:316)."""
    dev = _lib.require_gpu(x.device)
    x = x.to(torch.float32).contiguous()
    y = torch.empty_like(x)
    if x.shape[0]:
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().effocr_l2_normalize(_lib.ptr(x), x.shape[0], x.shape[1], _lib.ptr(y),
                                                      _lib.current_stream(dev)), "effocr_l2_normalize")
    return y
