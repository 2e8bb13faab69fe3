"""effocr_amd — MI355X (gfx950) implementation of EffOCR's recognizer hot path:
glyph crops -> encoder forward -> L2 normalise -> exact inner-product top-k -> glyph ids / chars,
behind the reference's own ``EffRecognizer`` / ``AutoEncoderFactory`` / ``FaissKNN`` call
conventions.  All compute is hand-written HIP in ``libeffocr_hip.so`` (C ABI: include/effocr_hip.h).
"""
from ._lib import EffOCRHipError, build, lib  # noqa: F401
from .weights import init_state_dict, load_checkpoint, save_checkpoint, embed_dim  # noqa: F401

__all__ = ["AutoEncoderFactory", "HipEncoder", "EffRecognizer", "EffLocalizer", "FaissKNN", "IndexFlatIP", "InferenceModel",
           "Recognizer", "build", "lib", "EffOCRHipError"]


def __getattr__(name):
    # lazy: the engine modules import torch.cuda-facing helpers only when used
    if name in ("AutoEncoderFactory", "HipEncoder"):
        from . import encoders
        return getattr(encoders, name)
    if name == "EffRecognizer":
        from .recognizer_engine import EffRecognizer
        return EffRecognizer
    if name in ("EffLocalizer", "HipLocalizer"):
        from . import localizer_engine
        return getattr(localizer_engine, name)
    if name in ("FaissKNN", "IndexFlatIP", "InferenceModel", "read_index", "write_index", "l2_normalize"):
        from . import knn
        return getattr(knn, name)
    if name in ("Recognizer", "create_batches", "iteration"):
        from . import pipeline
        return getattr(pipeline, name)
    raise AttributeError(name)
