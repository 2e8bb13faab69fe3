"""``EffRecognizer`` — drop-in for onnx_engines/recognizer_engine.py:6-27 on MI355X.

The reference wraps an ONNXRuntime session over ``enc_best.onnx`` (exported by
scripts/recognizer_onnx_export.py:63-69 with input ``imgs`` [B,3,224,224] and output ``embs``
[B,D]) and is called as ``recognizer_engine.run(batch)`` / ``recognizer_engine(batch)`` from
several Python threads sharing one instance (infer_effocr_onnx_multi.py:161-163,207-223,491-494).
``iteration`` returns ``(output, output)`` and the consumer reads ``embedding[0][0]`` (:371), so
``run`` must return a LIST whose element 0 is the ``[B,D]`` float32 ndarray.

Here ``model`` is the path of the encoder weights (``enc_best.pth`` state dict with ``net.`` keys, or
``.safetensors``) instead of an ``.onnx`` graph; there is no ONNXRuntime and no CPU fallback.
"""
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import weights as W
from .encoders import DEFAULT_PRECISION, HipEncoder


class _Lane:
    """One in-flight ``run`` call: its own HIP stream and pinned host staging buffers (grown on demand), so that the
    host->device copy of one call overlaps the kernels of another (the encoder keeps one workspace per stream)."""

    def __init__(self, device):
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        self.h_in = None
        self.h_out = None

    def staging(self, n_in, n_out):
        if self.h_in is None or self.h_in.numel() < n_in:
            self.h_in = torch.empty(n_in, dtype=torch.float32).pin_memory()
        if self.h_out is None or self.h_out.numel() < n_out:
            self.h_out = torch.empty(n_out, dtype=torch.float32).pin_memory()
        return self.h_in[:n_in], self.h_out[:n_out]


class EffRecognizer:

    def __init__(self, model, num_cores=None, providers=None, arch=None, precision=DEFAULT_PRECISION, img_size=224,
                 device=None, lanes=4, copiers=4, slices=8, staging="direct"):
        # num_cores / providers are ORT knobs (recognizer_engine.py:10-15): accepted and ignored.
        self.num_cores, self.providers = num_cores, providers
        if isinstance(model, dict):
            sd = W.strip_prefix(model)
        else:
            sd = W.load_checkpoint(model)
        self.arch = arch or W.infer_arch(sd)
        self._eng_net = HipEncoder(self.arch, sd, img_size=img_size, precision=precision, device=device)
        # One instance is shared by N Python threads in the reference (infer_effocr_onnx_multi.py:207-223,350-364).
        # `lanes` calls can be in flight at once; further callers wait for a free lane.  ctypes releases the GIL during
        # the enqueue and torch releases it during copies / synchronisation, so the threads really overlap:
        # H2D of call i+1 (pinned, async, own stream) runs under the kernels of call i.
        self._lanes = queue.LifoQueue()                    # LIFO: a single caller keeps re-using ONE lane (warm pinned buffers / workspace); N callers spread over N lanes
        for _ in range(max(1, int(lanes))):
            self._lanes.put(_Lane(self._eng_net.device))
        # staging="pinned" (rounds 3-5, kept for A/B): the pageable -> pinned staging copy is the slowest leg of such a call (38.5 MB per
        # 64 crops at one core's memcpy rate): a few persistent helper threads copy slices side by side (numpy releases the GIL for large
        # copies; torch's own intra-op pool is far too large on these hosts — see run()).  The default, "direct", needs none of it.
        self._copiers = ThreadPoolExecutor(max_workers=max(1, int(copiers)), thread_name_prefix="effocr-stage") if staging == "pinned" else None
        self._slices = max(1, int(slices))
        if staging not in ("direct", "pinned"):
            raise ValueError("staging must be 'direct' or 'pinned'")
        self._staging = staging

    def __call__(self, imgs):
        return self.run(imgs)

    @property
    def device(self):
        return self._eng_net.device

    @property
    def img_size(self):
        return self._eng_net.img_size

    @property
    def crop_dtype(self):
        """What a device-side crop producer should hand to ``encode_device`` (HipEncoder.crop_dtype): the encoder's 16-bit operand
        type for the ViTs in fp16 / bf16 mode, float32 otherwise."""
        return self._eng_net.crop_dtype

    def check_status(self):
        self._eng_net.check_status()

    def encode_device(self, x, normalize=False, chunk=2048):
        """Device-resident twin of ``run`` for callers whose crops are already in HBM (effocr_amd.pipeline.run_effocr):
        x [B,3,S,S] float32 (or ``crop_dtype``) on the engine's device -> [B,D] float32 on the device, asynchronous on the current
        stream.  Calls of more than ``chunk`` crops are encoded in slices (bounds the activation workspace: ~1.6 MB per ViT-S
        crop); the slices share one workspace and its STICKY status word, so the caller's one check_status() sees an overflow in
        any of them."""
        if x.shape[0] <= chunk:
            return self._eng_net.forward(x, normalize=normalize)
        return torch.cat([self._eng_net.forward(x[i:i + chunk], normalize=normalize) for i in range(0, x.shape[0], chunk)])

    def run(self, imgs):
        """imgs: np.ndarray[B,3,H,W] float32 -> [np.ndarray[B,D] float32] (ORT ``session.run`` shape)."""
        if isinstance(imgs, torch.Tensor):
            imgs = imgs.detach().cpu().numpy()
        if not isinstance(imgs, np.ndarray):
            raise ValueError("EffRecognizer.run expects a numpy array named 'imgs'")
        if imgs.dtype != np.float32:
            # ORT raises INVALID_ARGUMENT "Unexpected input data type" here
            raise ValueError(f"Unexpected input data type. Actual: {imgs.dtype}, expected: float32")
        if imgs.ndim != 4 or imgs.shape[1] != 3:
            raise ValueError(f"Invalid rank / channels for input: imgs, got shape {imgs.shape}")
        eng = self._eng_net
        B, D = int(imgs.shape[0]), eng.embed_dim
        if B == 0:
            return [np.empty((0, D), dtype=np.float32)]
        lane = self._lanes.get()                             # blocks while every lane is busy
        try:
            direct = self._staging == "direct"
            h_in, h_out = lane.staging(0 if direct else imgs.size, B * D)
            imgs = np.ascontiguousarray(imgs)
            if not direct:
                stage = h_in.view(imgs.shape)
                stage_np = stage.numpy()
            with torch.cuda.device(eng.device), torch.cuda.stream(lane.stream):
                # pageable -> pinned -> device in a few slices: the host memcpy of slice i+1 runs under the DMA of slice i.
                # (np.copyto, not Tensor.copy_: torch spreads a 38 MB copy over its whole intra-op pool — 128 threads on this
                # host — and the pool's wake-up costs 80-90 ms every few calls: 2.5 ms median but 23 ms mean per 64 crops.)
                x = torch.empty(imgs.shape, dtype=torch.float32, device=eng.device)
                if direct:
                    # round 6 (tools/host_register_probe.py): the runtime's own pageable -> device copy moves a 38.5 MB batch in 0.79 ms
                    # (48.9 GB/s; pinned -> device 0.69 ms) — the hand-made pageable -> pinned -> device chain below costs 1.5 ms on one
                    # core and 2.1-2.7 ms per call with its helper threads.  The copy blocks this caller (GIL released) on the lane's
                    # stream; other callers' kernels run meanwhile on theirs.
                    x.copy_(torch.from_numpy(imgs))
                    nsl = 0
                else:
                    nsl = max(1, min(self._slices, B // 8))
                bounds = [(B * i // nsl, B * (i + 1) // nsl) for i in range(nsl)]
                futs = [self._copiers.submit(np.copyto, stage_np[a:b], imgs[a:b]) for a, b in bounds]
                for (a, b), f in zip(bounds, futs):             # DMA of slice i as soon as its memcpy is done, in order
                    f.result()
                    x[a:b].copy_(stage[a:b], non_blocking=True)
                emb = eng.forward(x, normalize=False)
                h_out.view(B, D).copy_(emb, non_blocking=True)
                lane.stream.synchronize()
                out = h_out.view(B, D).numpy().copy()           # fresh ndarray owned by the caller
                if not np.isfinite(out).all():                  # f16 operand overflow / non-finite input: EFFOCR_EOVERFLOW, never silent
                    eng.check_status()
                    raise ValueError("EffRecognizer.run: non-finite embedding")
        finally:
            self._lanes.put(lane)
        return [out]
