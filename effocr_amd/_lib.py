"""ctypes binding of libeffocr_hip.so (include/effocr_hip.h) — the only door to the HIP kernels.

There is deliberately no fallback: if the shared library is missing or a call fails, an exception
is raised.  Nothing here (or anywhere in ``effocr_amd``) imports ``oracle/``.
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# EFFOCR_HIP_LIB: A/B experiments only (tools/): another build of the SAME library; the product default is the in-tree .so
SO_PATH = os.environ.get("EFFOCR_HIP_LIB") or os.path.join(_HERE, "libeffocr_hip.so")
# the product library + the kernels only A/B switches reach (row-panel GEMM, fused MLP without the projection phase): `make AB=1`.
# Never loaded by the engines on their own; tests that compare those paths switch to it with use_library().
SO_PATH_AB = os.path.join(_HERE, "libeffocr_hip_ab.so")

ABI_VERSION = 9          # == EFFOCR_ABI_VERSION of include/effocr_hip.h (tests/test_cabi.py checks the pair)
PREC = {"bf16": 0, "fp16": 1, "fp32": 2}
EPI = {"bias": 0, "bias_gelu": 1, "bias_resid": 2}

_lock = threading.Lock()
_lib = None
_loaded = {}             # path -> handle (use_library switches between them)


class EffOCRHipError(RuntimeError):
    pass


def build(force=False, verbose=False, ab=True):
    """Compile every HIP source for gfx950 into effocr_amd/libeffocr_hip.so (hipcc cross-compiles without a GPU) and, with ``ab``,
    the A/B build effocr_amd/libeffocr_hip_ab.so the tests of the alternative kernel paths load.  Returns the product library's path."""
    so = os.path.join(_HERE, "libeffocr_hip.so")
    for extra, target in (([], so),) + (((["AB=1"], SO_PATH_AB),) if ab else ()):
        cmd = ["make", "-C", _CSRC, "-j", str(min(16, os.cpu_count() or 1))] + extra
        if force and not extra:                                # the AB link reuses the objects the first pass just rebuilt
            cmd.append("-B")
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if verbose or res.returncode != 0:
            print(res.stdout)
        if res.returncode != 0 or not os.path.exists(target):
            raise EffOCRHipError(f"building {os.path.basename(target)} failed:\n" + res.stdout[-4000:])
    return so


def use_library(path=None):
    """Switch the process to another build of the library (tests: SO_PATH_AB) — or back to the default with ``None``.  Engine objects
    keep the handle they were created with, so objects made before and after a switch do not mix.  Returns the previous path."""
    global _lib, SO_PATH, EXPORTS
    with _lock:
        prev = SO_PATH
        SO_PATH = path or os.environ.get("EFFOCR_HIP_LIB") or os.path.join(_HERE, "libeffocr_hip.so")
        if SO_PATH != prev:
            _lib = _loaded.get(SO_PATH)
            if _lib is not None:
                EXPORTS = sorted(_declare(_lib).keys())
    return prev


def _declare(lib):
    c = ctypes
    vp, i32, i64, sz, f32p, i64p = c.c_void_p, c.c_int, c.c_int64, c.c_size_t, c.c_void_p, c.c_void_p
    sig = {
        "effocr_abi_version": (i32, []),
        "effocr_last_error": (c.c_char_p, []),
        "effocr_encoder_create": (i32, [c.c_char_p, i32, i32, c.POINTER(vp)]),
        "effocr_encoder_destroy": (None, [vp]),
        "effocr_encoder_embed_dim": (i32, [vp]),
        "effocr_encoder_num_params": (i32, [vp]),
        "effocr_encoder_param_name": (c.c_char_p, [vp, i32]),
        "effocr_encoder_param_numel": (i64, [vp, i32]),
        "effocr_encoder_set_param": (i32, [vp, c.c_char_p, vp, i64]),
        "effocr_encoder_weights_bytes": (sz, [vp]),
        "effocr_encoder_upload": (i32, [vp, vp, sz]),
        "effocr_encoder_workspace_bytes": (sz, [vp, i32]),
        "effocr_encoder_forward": (i32, [vp, f32p, i32, f32p, i32, vp, sz, vp]),
        "effocr_encoder_forward_ex": (i32, [vp, vp, i32, i32, f32p, i32, vp, sz, vp]),
        "effocr_encoder_check_status": (i32, [vp, vp, vp]),
        "effocr_encoder_reset_status": (i32, [vp, vp, vp]),
        "effocr_clock_sample": (i32, [vp, vp]),
        "effocr_encoder_set_chunk": (i32, [vp, i32]),
        "effocr_encoder_set_option": (i32, [vp, c.c_char_p, i32]),
        "effocr_op_ln_linear": (i32, [i32, i32, f32p, f32p, f32p, c.c_float, vp, f32p, f32p, vp, i32, i32, i32, vp]),
        "effocr_encoder_profile_begin": (i32, [vp, i32, c.c_char_p]),
        "effocr_encoder_profile_collect": (i32, [vp]),
        "effocr_encoder_profile_get": (i32, [vp, i32, c.POINTER(c.c_char_p), c.POINTER(c.c_double), c.POINTER(i32),
                                             c.POINTER(c.c_double)]),
        "effocr_encoder_profile_clock": (i32, [vp, i32, c.POINTER(c.c_double)]),
        "effocr_knn_workspace_bytes": (sz, [i64, i64, i32, i32]),
        "effocr_knn_ip_topk": (i32, [f32p, i64, f32p, i64, i32, i32, f32p, i64p, vp, sz, vp]),
        "effocr_knn_screen_workspace_bytes": (sz, [i64, i64, i32, i32]),
        "effocr_knn_set_option": (i32, [c.c_char_p, i32]),
        "effocr_knn_screen_flag_offset": (sz, [i64, i64, i32, i32]),
        "effocr_knn_ip_topk_screened": (i32, [f32p, i64, f32p, vp, i64, i32, i32, c.c_float, f32p, i64p, vp, sz, vp]),
        "effocr_convert_bf16": (i32, [f32p, i64, vp, vp]),
        "effocr_bf16_blocked_bytes": (sz, [i64, i32]),
        "effocr_convert_bf16_blocked": (i32, [f32p, i64, i32, vp, vp]),
        "effocr_knn_ip_topk_screened2": (i32, [f32p, i64, f32p, vp, vp, i64, i32, i32, c.c_float, f32p, i64p, vp, sz, vp]),
        "effocr_l2_normalize": (i32, [f32p, i64, i32, f32p, vp]),
        "effocr_gather_rows": (i32, [f32p, i64p, i64, i32, f32p, vp]),
        "effocr_crop_transform": (i32, [vp, i32, i32, i64, vp, i32, i32, i32, c.POINTER(c.c_float), c.POINTER(c.c_float),
                                        c.POINTER(c.c_float), f32p, vp]),
        "effocr_crop_transform_batch": (i32, [vp, i32, i64, i32, i32, i64, vp, i64, i32, i32, c.POINTER(c.c_float), c.POINTER(c.c_float),
                                              c.POINTER(c.c_float), f32p, vp]),
        "effocr_crop_transform_batch_ex": (i32, [vp, i32, i64, i32, i32, i64, vp, i64, i32, i32, c.POINTER(c.c_float), c.POINTER(c.c_float),
                                                 c.POINTER(c.c_float), i32, vp, vp]),
        "effocr_localizer_create": (i32, [c.c_char_p, i32, i32, i32, c.POINTER(vp)]),
        "effocr_localizer_destroy": (None, [vp]),
        "effocr_localizer_num_params": (i32, [vp]),
        "effocr_localizer_param_name": (c.c_char_p, [vp, i32]),
        "effocr_localizer_param_numel": (i64, [vp, i32]),
        "effocr_localizer_set_param": (i32, [vp, c.c_char_p, vp, i64]),
        "effocr_localizer_weights_bytes": (sz, [vp]),
        "effocr_localizer_upload": (i32, [vp, vp, sz]),
        "effocr_localizer_set_option": (i32, [vp, c.c_char_p, i32]),
        "effocr_localizer_num_predictions": (i64, [vp]),
        "effocr_localizer_num_outputs": (i32, [vp]),
        "effocr_localizer_workspace_bytes": (sz, [vp, i32]),
        "effocr_localizer_forward": (i32, [vp, f32p, i32, f32p, vp, sz, vp]),
        "effocr_letterbox": (i32, [vp, i32, i32, i64, i32, i32, i32, i32, i32, i32, i32, f32p, vp]),
        "effocr_nms_workspace_bytes": (sz, [i32, i32]),
        "effocr_nms_batch_workspace_bytes": (sz, [i32, i32, i32]),
        "effocr_nms": (i32, [f32p, i32, i32, c.c_float, c.c_float, i32, i32, c.c_float, i32, f32p, vp, vp, sz, vp]),
        "effocr_nms_batch": (i32, [f32p, i32, i32, i32, c.c_float, c.c_float, i32, i32, c.c_float, i32, f32p, vp, vp, sz, vp]),
        "effocr_parse_char_boxes": (i32, [f32p, vp, i32, i32, i32, i32, i32, i32, f32p, vp, vp, vp, vp]),
        "effocr_op_linear": (i32, [i32, i32, vp, vp, f32p, f32p, vp, i32, i32, i32, vp]),
        "effocr_op_layernorm": (i32, [i32, f32p, i64, i32, f32p, f32p, c.c_float, vp, vp]),
        "effocr_op_attention": (i32, [i32, vp, vp, i32, i32, i32, vp]),
        "effocr_op_linear_blocked": (i32, [i32, i32, vp, vp, f32p, f32p, vp, i32, i32, i32, i32, vp]),
        "effocr_op_mlp_blocked": (i32, [i32, f32p, f32p, f32p, c.c_float, vp, f32p, vp, f32p, f32p, i32, i32, i32, i32, vp, sz, vp]),
        "effocr_op_mlp_ln_blocked": (i32, [i32, f32p, f32p, f32p, c.c_float, vp, f32p, vp, f32p, f32p, f32p, f32p, vp, i32, i32, i32, i32, vp, sz, vp]),
        "effocr_op_proj_mlp_blocked": (i32, [i32, f32p, vp, vp, f32p, f32p, f32p, c.c_float, vp, f32p, vp, f32p, f32p, i32, i32, i32, i32, vp, sz, vp]),
        "effocr_op_qkv_attn_blocked": (i32, [i32, vp, vp, f32p, vp, i32, i32, i32, i32, vp]),
        "effocr_op_layernorm_blocked": (i32, [i32, f32p, i64, i32, f32p, f32p, c.c_float, vp, vp]),
    }
    for name, (res, args) in sig.items():
        if os.environ.get("EFFOCR_HIP_LIB") and not hasattr(lib, name):
            # A/B against an older build of the SAME ABI (tools/ab_bench.sh): a symbol added since is absent — calling it must fail
            # loudly, never run with ctypes' default int signature
            def _absent(*a, _n=name, **k):
                raise EffOCRHipError(f"{_n} is not exported by the library EFFOCR_HIP_LIB points at")
            setattr(lib, name, _absent)
            continue
        fn = getattr(lib, name)            # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return sig


EXPORTS = None


def lib():
    """Load (once) and return the ctypes handle; raises if the extension has not been built."""
    global _lib, EXPORTS
    with _lock:
        if _lib is None:
            if not os.path.exists(SO_PATH):
                raise EffOCRHipError(
                    f"{SO_PATH} not found: the HIP extension is required (no CPU fallback). "
                    "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C effocr_amd/csrc`.")
            handle = ctypes.CDLL(SO_PATH)
            EXPORTS = sorted(_declare(handle).keys())
            # the override keeps the ABI check: an older-or-equal version only behind a second, explicit flag (A/B runs against the
            # previous round's build), and never a newer or foreign one
            got = handle.effocr_abi_version()
            older_ok = bool(os.environ.get("EFFOCR_HIP_LIB")) and os.environ.get("EFFOCR_HIP_ALLOW_OLDER_ABI") == "1" and 0 < got <= ABI_VERSION
            if got != ABI_VERSION and not older_ok:
                raise EffOCRHipError(f"libeffocr_hip.so ABI version {got} != {ABI_VERSION} expected by this package: rebuild (make -C effocr_amd/csrc)")
            _lib = _loaded[SO_PATH] = handle
    return _lib


def check(rc, what="", handle=None):
    """``handle``: the library the failing call was made through (engines pass the one they were created with — last_error is
    per shared object, and use_library() may have switched the process default since)."""
    if rc != 0:
        msg = (handle or lib()).effocr_last_error()
        raise EffOCRHipError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(device=None):
    """Resolve ``device`` to a concrete GPU.  ``None`` and an index-less ``"cuda"`` mean torch's CURRENT device — the semantics
    of the reference's default ``--device cuda`` (infer_effocr.py:439,178) — so that with one process per GPU
    (``torch.cuda.set_device(LOCAL_RANK)``) every engine of a rank lands on that rank's GPU, never on a literal GPU 0."""
    import torch
    if not torch.cuda.is_available():
        raise EffOCRHipError("no ROCm GPU visible: the EffOCR HIP path has no CPU fallback")
    d = torch.device("cuda" if device is None else device)
    if d.type != "cuda":
        raise EffOCRHipError(f"device {device!r} is not a GPU: the EffOCR HIP path has no CPU fallback")
    if d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return d
