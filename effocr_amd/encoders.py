"""Recognizer encoder engine on MI355X, behind the reference's two call conventions.

* ``AutoEncoderFactory(backend, modelpath)`` mirrors models/encoders.py:50-97: it returns a class
  whose instances behave like the reference's ``AutoEncoder`` ``nn.Module`` at its call sites —
  ``encoder.load(ckpt)`` (infer_effocr.py:177), ``.to(device)`` / ``.eval()`` (:178-179),
  ``encoder(x[B,3,H,W]) -> [B,D]`` (:314), ``.named_parameters()`` (:538-540).
* ``EffRecognizer`` (effocr_amd/recognizer_engine.py) keeps onnx_engines/recognizer_engine.py:6-27.

Both sit on ``HipEncoder``, a thin owner of the C-ABI encoder handle (include/effocr_hip.h):
weights are packed by the library and live in a device blob allocated by torch; the forward pass is
a fixed sequence of hand-written gfx950 kernels launched on torch's current HIP stream.
"""
import ctypes
import threading
import warnings

import numpy as np
import torch

from . import _lib
from . import weights as W

# Operand precision the engines take when the caller does not say.  north_star's bar is "encoder embeddings within 1e-3 rel fp32"
# (the reference computes in fp32: infer_effocr.py:314-316; the only tolerance it states itself is the ONNX export's,
# scripts/recognizer_onnx_export.py:81,84): "fp16" meets it (7.5e-4, f16 MFMA operands, fp32 accumulation / residual / LayerNorm /
# softmax; overflow of f16's range is detected and reported as EFFOCR_EOVERFLOW), "bf16" does not (5.7e-3; top-1 ids identical
# wherever two glyphs are more than 1e-4 apart in cosine) and stays one keyword away for users who want its 4 % higher rate and
# fp32's exponent range; "fp32" is the exact parity mode.  bench.py's headline keeps bf16 because BASELINE.json names it.
DEFAULT_PRECISION = "fp16"
_CROP_DTYPE = {"fp16": torch.float16, "bf16": torch.bfloat16}


class HipEncoder:
    """Device-resident encoder: C-ABI handle + weight blob + one grow-only workspace PER HIP STREAM (the Python lock
    covers only the enqueue; two threads forwarding on different streams must not share activation buffers)."""

    def __init__(self, arch, state_dict, img_size=224, precision=DEFAULT_PRECISION, device=None):
        if precision not in _lib.PREC:
            raise ValueError(f"precision must be one of {sorted(_lib.PREC)}, got {precision!r}")
        self.device = _lib.require_gpu(device)
        self.arch, self.img_size, self.precision = arch, int(img_size), precision
        self._L = _lib.lib()
        self._lock = threading.Lock()
        self._h = ctypes.c_void_p()
        _lib.check(self._L.effocr_encoder_create(arch.encode(), self.img_size, _lib.PREC[precision],
                                                 ctypes.byref(self._h)), "effocr_encoder_create", self._L)
        self.embed_dim = int(self._L.effocr_encoder_embed_dim(self._h))
        sd = W.strip_prefix(state_dict)
        W.check_state_dict(arch, sd, self.img_size)
        for i in range(self._L.effocr_encoder_num_params(self._h)):
            name = self._L.effocr_encoder_param_name(self._h, i).decode()
            t = sd[name].detach().to("cpu", torch.float32).contiguous()
            _lib.check(self._L.effocr_encoder_set_param(self._h, name.encode(), _lib.ptr(t), t.numel()),
                       f"effocr_encoder_set_param({name})", self._L)
        nbytes = int(self._L.effocr_encoder_weights_bytes(self._h))
        with torch.cuda.device(self.device):
            self._wblob = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            _lib.check(self._L.effocr_encoder_upload(self._h, _lib.ptr(self._wblob), nbytes), "effocr_encoder_upload", self._L)
        self._ws = {}
        # calls of 192..640 crops run as 2-4 concurrent sub-batches of ~128 crops on side streams (ViT-S, 16-bit modes; forward_split below)
        self.split_streams = True
        self._side = None                    # (streams, thread pool) of the split, created on first use
        self._side_used = set()              # side streams whose workspaces may hold an unchecked status word
        self._stream_locks = {}              # stream handle -> lock held while a forward is enqueued onto that stream
        self._profiling = False

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self._L.effocr_encoder_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    def set_chunk(self, crops_per_chunk):
        """Internal sub-batch size of the ViT forward (0 = whole batch); see effocr_encoder_set_chunk."""
        _lib.check(self._L.effocr_encoder_set_chunk(self._h, int(crops_per_chunk)), "effocr_encoder_set_chunk", self._L)
        # (the per-stream workspaces stay: they are grow-only scratch, re-sized by the next forward if it needs more, and their first
        # word is the sticky status — dropping them would drop an unchecked overflow)

    def set_option(self, name, value):
        _lib.check(self._L.effocr_encoder_set_option(self._h, name.encode(), int(value)), "effocr_encoder_set_option", self._L)

    def workspace_bytes(self, batch):
        return int(self._L.effocr_encoder_workspace_bytes(self._h, int(batch)))

    @property
    def crop_dtype(self):
        """Element type a crop producer on the device should hand over (PairedTransform.boxes_batch(dtype=...), SURVEY f-2): the
        ViT encoders' own 16-bit operand type — the patch embedding rounds fp32 crops to it anyway, so the embeddings are
        bit-identical and half the bytes move — float32 for the fp32 mode and the CNN."""
        if self.arch.startswith("vit") and self.precision in _CROP_DTYPE:
            return _CROP_DTYPE[self.precision]
        return torch.float32

    def forward(self, x, normalize=False):
        """x: [B,3,H,W] CUDA tensor, float32 (the reference's type) or ``crop_dtype`` -> [B,D] float32 CUDA tensor (async on the
        current stream)."""
        if not isinstance(x, torch.Tensor):
            raise TypeError("HipEncoder.forward expects a torch.Tensor")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.img_size or x.shape[3] != self.img_size:
            raise ValueError(f"expected input [B,3,{self.img_size},{self.img_size}], got {tuple(x.shape)}")
        if x.dtype != torch.float32 and x.dtype != self.crop_dtype:
            raise ValueError(f"expected float32 input{'' if self.crop_dtype == torch.float32 else f' (or {self.crop_dtype})'}, got {x.dtype}")
        if x.device != self.device:
            raise ValueError(f"input is on {x.device}, encoder on {self.device}")
        x = x.contiguous()
        B = x.shape[0]
        emb = torch.empty((B, self.embed_dim), dtype=torch.float32, device=self.device)
        if B == 0:
            return emb
        parts = self._split_plan(B)
        if parts > 1:
            return self._forward_split(x, emb, normalize, parts)
        self._forward_into(x, emb, normalize)
        return emb

    # Measured (tools/split_streams.py, tools/split_sweep.py, round 6, same box, encoder + k-NN): a 256-crop call 78.9 k crops/s as ONE call
    # against 87.1 / 89.5 k as two / three concurrent sub-batches (0.83 -> 0.94 of the 1024-crop rate), 192 crops 66.4 -> 87.3 k as two,
    # 384 crops 83.5 -> 91.3 k as three, 512 crops 90.9 -> 93.5 k; 160 crops and below lose (72.3 -> 59.4 k), 640 and above do not gain.  Why: 256 crops = 394 fused-MLP
    # panels = 1.54 rounds of 256 CUs — the second round runs on 54 % of the chip — while two 128-crop sub-batches out of phase fill each
    # other's idle CUs (one's 197-panel MLP beside the other's per-image kernel).
    # Re-measured with the 64-token pair panels in the library (round 6, second half; gpurun_out/r6d_split_sweep.txt): 168 crops 66.7 -> 76.4 k as
    # two, 240 / 249 / 256 crops 89.6 / 92.8 / 92.0 k as two against 75.1 / 75.6 / 89.9 k as three (sub-batches of 80-85 crops are pair-panel
    # calls now: good alone, but their 160-170 workgroups leave the other sub-batches no idle CUs), 288 crops 89.1 (two) / 92.6 k (three).
    # Third pass, with the FINAL pair panels (projection / LayerNorm / stores split between the waves of a pair: 30-83-crop sub-batches cost 0.83-0.88 of
    # what they did; profiles/r06d_split_sweep.txt, k crops/s as 1 / 2 / 3 parts): 84: 64.1 / 59.2 / 47.7; 90: 63.7 / 67.1 / 67.8; 96: 67.5 / 71.3 / 68.2;
    # 104: 72.4 / 75.1 / 69.9; 112: 76.0 / 68.1 / 70.6; 120: 79.3 / 72.2 / 71.8; 128: 82.2 / 74.5 / 73.4; 136: 69.5 / 75.4 / 76.7; 144: 71.4 / 74.4 / 77.8;
    # 152: 73.3 / 76.4 / 80.0; 160: 74.5 / 77.9 / 76.6; 168: 64.7 / 76.2 / 78.8; 176: 66.6 / 80.9 / 78.0; 184: 68.0 / 85.3 / 78.7; 256: 83.3 / 90.7 / 92.8 (noise:
    # 92.0 / 89.9 in the second pass); 272: 83.1 / 84.7 / 94.1.  Above 128 images the per-image kernel loses its head split, below ~110 crops the
    # 128-token panels fill less than 2/3 of the CUs: in both ranges two or three pair-panel sub-batches side by side win 5-10 %.
    SPLIT_MIN, SPLIT_MAX, SPLIT_THREE = 88, 576, 272
    SPLIT_SINGLE = (108, 132)                                # ... except here: one call (112-128 crops: 172-197 panels, heads split 2-way, fill the chip best)
    SPLIT_THREE_MID = (132, 160)                             # three sub-batches of 44-53 crops

    def _split_plan(self, B):
        if not (self.split_streams and self.arch in ("vit_small_patch16_224", "vit_base_patch16_224") and self.precision in _CROP_DTYPE) or self._profiling:
            return 1
        if isinstance(self.split_streams, int) and not isinstance(self.split_streams, bool):
            return max(1, min(4, self.split_streams)) if B >= 8 else 1        # (forced part count: tools/split_sweep.py)
        if self.arch == "vit_base_patch16_224":
            # tools/split_sweep_vitb.py, three interleaved rounds: 256 crops 24.6 -> 25.6 k crops/s as two concurrent sub-batches, 512 crops
            # 26.1 -> 26.8 k, 1024 crops 26.85 -> 27.3 k (three parts lose): the HBM-bound attention / residual epilogues of one beside the
            # matrix-bound linears of the other
            return 2 if B >= 192 else 1
        if not (self.SPLIT_MIN <= B < self.SPLIT_MAX) or self.SPLIT_SINGLE[0] <= B < self.SPLIT_SINGLE[1]:
            return 1
        if self.SPLIT_THREE_MID[0] <= B < self.SPLIT_THREE_MID[1]:
            return 3
        return 2 if B < self.SPLIT_THREE else 3                               # tools/split_sweep.py (profiles/r06d_split_sweep.txt)

    def _forward_into(self, x, emb, normalize):
        """Enqueue one forward on torch's current stream of THIS thread.  The engine lock covers the workspace table only: the library's
        forward is re-entrant while its profiler is not armed (it reads the handle, writes only the caller's buffers), and N caller
        threads / the split's helper threads enqueue side by side on DIFFERENT streams (ctypes releases the GIL); forwards onto the same
        stream are serialised by that stream's lock, an armed profiler serialises everything."""
        B = x.shape[0]
        need = self.workspace_bytes(B)
        with torch.cuda.device(self.device):
            with self._lock:
                key = torch.cuda.current_stream(self.device).cuda_stream
                ws = self._ws.get(key)
                if ws is None or ws.numel() < need:
                    old = self._ws.pop(key, None)
                    ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                    if old is None:
                        ws[:256].zero_()                            # the sticky status word starts clean (effocr_encoder_check_status)
                    else:
                        ws[:256].copy_(old[:256])                   # ... and SURVIVES a larger workspace: an overflow recorded by an earlier, not yet
                    self._ws[key] = ws                              # checked forward on this stream must still be reported by the next check
                slock = self._stream_locks.get(key)
                if slock is None:
                    slock = self._stream_locks[key] = threading.Lock()
                serial = self._profiling
                if serial:
                    self._enqueue(x, emb, normalize, ws)
            if not serial:
                # one forward at a time PER STREAM: its kernels share that stream's workspace, and two threads enqueueing onto one stream
                # (two callers' sub-batches on a shared side stream) would interleave their launch sequences over the same activations
                with slock:
                    self._enqueue(x, emb, normalize, ws)

    def _enqueue(self, x, emb, normalize, ws):
        x_dtype = _lib.PREC["fp32"] if x.dtype == torch.float32 else _lib.PREC[self.precision]
        _lib.check(self._L.effocr_encoder_forward_ex(self._h, _lib.ptr(x), x_dtype, x.shape[0], _lib.ptr(emb), 1 if normalize else 0,
                                                     _lib.ptr(ws), ws.numel(),
                                                     _lib.current_stream(self.device)), "effocr_encoder_forward", self._L)

    def _forward_split(self, x, emb, normalize, parts):
        from concurrent.futures import ThreadPoolExecutor
        with self._lock:
            if self._side is None:
                self._side = ([torch.cuda.Stream(device=self.device) for _ in range(4)],
                              ThreadPoolExecutor(max_workers=4, thread_name_prefix="effocr-split"))
        streams, pool = self._side
        cur = torch.cuda.current_stream(self.device)
        B = x.shape[0]
        ready = torch.cuda.Event()
        ready.record(cur)                                      # the crops (and everything queued before them) are on `cur`
        bounds = [(B * i // parts, B * (i + 1) // parts) for i in range(parts)]

        def sub(i):
            a, b = bounds[i]
            with torch.cuda.device(self.device), torch.cuda.stream(streams[i]):
                streams[i].wait_event(ready)
                self._forward_into(x[a:b], emb[a:b], normalize)
                done = torch.cuda.Event()
                done.record(streams[i])
            return done

        futs = [pool.submit(sub, i) for i in range(parts)]
        first_error = None
        for i, f in enumerate(futs):
            self._side_used.add(streams[i].cuda_stream)
            try:
                cur.wait_event(f.result())                     # join: `cur` continues (k-NN, the caller's reads) behind every sub-batch
            except Exception as e:                             # a failed enqueue: still join the others (they read x, write emb) before raising
                cur.wait_stream(streams[i])
                first_error = first_error or e
        if first_error is not None:
            raise first_error
        return emb

    def check_status(self):
        """Synchronise the current stream and raise EffOCRHipError (EFFOCR_EOVERFLOW) if ANY forward issued on it since the previous
        check produced a non-finite embedding — how an f16 operand overflow surfaces (include/effocr_hip.h; the status word is
        sticky and this call clears it).  Callers that synchronise anyway (EffRecognizer.run, Recognizer.__call__, run_effocr,
        ShardedRecognizer) call it there; a purely asynchronous user (HipEncoder.forward, the AutoEncoder twin, Recognizer.neighbors)
        calls it when it wants to know, and a non-finite query can never come back from the k-NN with a plausible id in the meantime
        (knn.hip ranks a NaN score below every real one: ids -1)."""
        with self._lock, torch.cuda.device(self.device):
            side, self._side_used = self._side_used, set()
            for key in side:                                   # sub-batches of split calls: their status words live in the side streams' workspaces
                ws = self._ws.get(key)
                if ws is not None:
                    _lib.check(self._L.effocr_encoder_check_status(self._h, _lib.ptr(ws), ctypes.c_void_p(key)),
                               "effocr_encoder_check_status", self._L)
            ws = self._ws.get(torch.cuda.current_stream(self.device).cuda_stream)
            if ws is None:
                return
            _lib.check(self._L.effocr_encoder_check_status(self._h, _lib.ptr(ws), _lib.current_stream(self.device)),
                       "effocr_encoder_check_status", self._L)

    __call__ = forward

    # -- HIP-event profiler (bench.py roofline) --------------------------------------------------
    def profile_begin(self, only=None):
        """Arm the in-library profiler: every kernel class, or only the class named ``only``."""
        self._profiling = True                                 # (armed profiler: enqueues serialised, no stream split)
        _lib.check(self._L.effocr_encoder_profile_begin(self._h, 2 if only else 1, only.encode() if only else None),
                   "effocr_encoder_profile_begin", self._L)

    def profile_collect(self):
        """-> {class: {"ms": total, "launches": n, "flops": total algorithmic FLOPs, "shader_ghz": clock the class ran at (full
        breakdowns only, else 0)}} (synchronises)."""
        n = self._L.effocr_encoder_profile_collect(self._h)
        self._profiling = False
        if n < 0:
            _lib.check(n, "effocr_encoder_profile_collect")
        out = {}
        for i in range(n):
            name, ms, cnt, work = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
            _lib.check(self._L.effocr_encoder_profile_get(self._h, i, ctypes.byref(name), ctypes.byref(ms),
                                                          ctypes.byref(cnt), ctypes.byref(work)), "profile_get", self._L)
            ghz = ctypes.c_double()
            _lib.check(self._L.effocr_encoder_profile_clock(self._h, i, ctypes.byref(ghz)), "profile_clock", self._L)
            out[name.value.decode()] = {"ms": ms.value, "launches": cnt.value, "flops": work.value, "shader_ghz": ghz.value}
        return out


def AutoEncoderFactory(backend, modelpath, precision=DEFAULT_PRECISION, img_size=224):
    """Drop-in for models/encoders.py:50 ``AutoEncoderFactory(backend, modelpath)``.

    Only the ``"timm"`` backend with the architectures BASELINE.json names is implemented (the "hf"
    branch and XcitDinoEncoder are out of scope); anything else raises
    NotImplementedError exactly like the reference's ``else`` branch (encoders.py:93-95).
    ``precision`` / ``img_size`` are extensions with reference-compatible defaults.
    """
    if backend != "timm":
        raise NotImplementedError
    W.embed_dim(modelpath)          # raises NotImplementedError for unknown architectures

    class AutoEncoder:
        arch = modelpath

        def __init__(self, model=modelpath, device="cuda", seed=0):
            # the reference downloads ImageNet weights here (pretrained=True, encoders.py:58); with
            # no network the instance starts from a seeded random init until load_state_dict().
            self.model_name = model
            self._sd = W.init_state_dict(model, seed=seed, img_size=img_size)
            self._device = self._resolve(device)
            self._engine = None
            self.training = False

        @staticmethod
        def _resolve(device):
            # an index-less "cuda" is torch's CURRENT device (what nn.Module.to("cuda") means at infer_effocr.py:178), resolved when
            # it is named: one process per GPU sets the device once, before building its engines
            d = torch.device("cuda" if device is None else device)
            if d.type == "cuda" and d.index is None and torch.cuda.is_available():
                d = torch.device("cuda", torch.cuda.current_device())
            return d

        # -- checkpoint I/O (encoders.py:66-70) ------------------------------------------------
        @classmethod
        def load(cls, checkpoint):
            ptnet = cls()
            ptnet.load_state_dict(W.load_checkpoint(checkpoint))
            return ptnet

        def load_state_dict(self, sd, strict=True):
            sd = W.strip_prefix(sd)
            W.check_state_dict(self.model_name, sd, img_size)
            self._sd = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in sd.items()}
            self._engine = None

        def state_dict(self):
            return {"net." + k: v for k, v in self._sd.items()}

        # -- nn.Module look-alikes used by infer_effocr.py:178-179,538-540 ---------------------
        def to(self, device):
            device = self._resolve(device)
            if device != self._device:
                self._device, self._engine = device, None
            return self

        def eval(self):
            self.training = False
            return self

        def named_parameters(self):
            shapes = W.param_shapes(self.model_name, img_size)
            for k, v in self._sd.items():
                if k in shapes and not k.endswith(("running_mean", "running_var")):
                    p = torch.nn.Parameter(v, requires_grad=True)
                    yield "net." + k, p

        def parameters(self):
            for _, p in self.named_parameters():
                yield p

        @property
        def engine(self):
            if self._engine is None:
                self._engine = HipEncoder(self.model_name, self._sd, img_size=img_size, precision=precision,
                                          device=self._device)
            return self._engine

        def forward(self, x):
            return self.engine.forward(x, normalize=False)

        def check_status(self):
            """Raise if any forward since the last check produced a non-finite embedding (HipEncoder.check_status)."""
            if self._engine is not None:
                self._engine.check_status()

        __call__ = forward

    AutoEncoder.__name__ = "AutoEncoder"
    return AutoEncoder
