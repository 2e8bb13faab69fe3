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
import threading

import numpy as np
import torch

from . import weights as W
from .encoders import HipEncoder


class EffRecognizer:

    def __init__(self, model, num_cores=None, providers=None, arch=None, precision="bf16", img_size=224,
                 device="cuda:0"):
        # num_cores / providers are ORT knobs (recognizer_engine.py:10-15): accepted and ignored.
        self.num_cores, self.providers = num_cores, providers
        if isinstance(model, dict):
            sd = W.strip_prefix(model)
        else:
            sd = W.load_checkpoint(model)
        self.arch = arch or W.infer_arch(sd)
        self._eng_net = HipEncoder(self.arch, sd, img_size=img_size, precision=precision, device=device)
        self._run_lock = threading.Lock()   # one instance is shared by N threads in the reference

    def __call__(self, imgs):
        return self.run(imgs)

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
        with self._run_lock:
            x = torch.from_numpy(np.ascontiguousarray(imgs)).to(self._eng_net.device, non_blocking=False)
            emb = self._eng_net.forward(x, normalize=False)
            out = emb.cpu().numpy()
        return [out]
