"""-m "not gpu": libeffocr_hip.so loads, exports every function include/effocr_hip.h declares, and
its host-side logic (parameter tables, sizes, argument checking, error strings) works without a GPU."""
import ctypes
import os
import re

import pytest

from effocr_amd import _lib
from effocr_amd.weights import param_shapes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "effocr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(effocr_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(hip_lib):
    names = declared_functions()
    assert len(names) >= 25
    raw = ctypes.CDLL(_lib.SO_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in effocr_hip.h but not exported"
    assert set(_lib.EXPORTS) <= set(names)
    header = open(os.path.join(ROOT, "include", "effocr_hip.h")).read()
    declared = int(re.search(r"#define\s+EFFOCR_ABI_VERSION\s+(\d+)", header).group(1))
    assert hip_lib.effocr_abi_version() == declared == _lib.ABI_VERSION


def test_product_and_ab_builds(hip_lib):
    """`make` = the product library (only what the default dispatch and the documented modes can reach: about 4 MB); `make AB=1` = the
    same + the row-panel GEMM and the fused MLP without the projection phase for the A/B tests.  Same exported ABI; in the product
    build the A/B-only entry points fail loudly with EFFOCR_EUNSUPPORTED instead of not existing."""
    assert os.path.getsize(os.path.join(ROOT, "effocr_amd", "libeffocr_hip.so")) < 7.2e6   # (round 6, second half: 5.76 MB, then 6.3 MB with the pair parts (3- and 6-way, four bodies each) — the 64-token pair kernel and the "partial sums start at zero" bodies of the split parts, two per 16-bit type; before: 4.0 MB + round 4's LayerNorm-folded gemm3 epilogues + round 5: 16-bit-input patch embedding, Q-stationary k-NN screen, persistent gemm3 tiles + round 6: the per-image kernel's one-tile body for the padded wave, the box stage, the two-chunk split parts of the fused MLP)
    assert os.path.exists(_lib.SO_PATH_AB), "build the A/B library: python -c 'import __graft_entry__ as g; g.build()'"
    assert os.path.getsize(_lib.SO_PATH_AB) > os.path.getsize(os.path.join(ROOT, "effocr_amd", "libeffocr_hip.so")) + 2e6
    ab = ctypes.CDLL(_lib.SO_PATH_AB)
    for n in declared_functions():
        assert hasattr(ab, n), f"{n} missing from the A/B build"
    assert ab.effocr_abi_version() == hip_lib.effocr_abi_version()
    # an A/B-only operator through the PRODUCT library: a 16-bit linear at a ViT-S width dispatches to the row-panel GEMM (no GPU needed:
    # the stub fails before any launch)
    assert hip_lib.effocr_op_linear(0, 0, ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), None, ctypes.c_void_p(16), 4, 384, 384, None) == -2
    assert b"A/B path" in hip_lib.effocr_last_error()
    prev = _lib.use_library(_lib.SO_PATH_AB)
    try:
        assert _lib.lib() is not hip_lib and _lib.SO_PATH == _lib.SO_PATH_AB
    finally:
        _lib.use_library(None if prev.endswith("libeffocr_hip.so") else prev)
    assert _lib.lib() is hip_lib


@pytest.mark.parametrize("arch,img,D", [("vit_small_patch16_224", 224, 384), ("vit_base_patch16_224", 224, 768),
                                        ("resnet18", 32, 512), ("vit_tiny_test", 64, 128)])
def test_encoder_handle_param_table(hip_lib, arch, img, D):
    L = hip_lib
    h = ctypes.c_void_p()
    assert L.effocr_encoder_create(arch.encode(), img, 0, ctypes.byref(h)) == 0
    try:
        assert L.effocr_encoder_embed_dim(h) == D
        want = param_shapes(arch, img)
        n = L.effocr_encoder_num_params(h)
        got = {L.effocr_encoder_param_name(h, i).decode(): L.effocr_encoder_param_numel(h, i) for i in range(n)}
        assert list(got) == list(want)                                   # same names, same order as the timm state dict
        for k, shp in want.items():
            numel = 1
            for d in shp:
                numel *= d
            assert got[k] == numel
        assert L.effocr_encoder_weights_bytes(h) > 0
        w1, w8 = L.effocr_encoder_workspace_bytes(h, 1), L.effocr_encoder_workspace_bytes(h, 8)
        assert 0 < w1 < w8
        # call-order and argument errors are reported, not crashed on
        assert L.effocr_encoder_set_param(h, b"no.such.param", None, 0) == -1
        import numpy as np
        buf = np.zeros(3, np.float32)
        assert L.effocr_encoder_set_param(h, b"norm.weight" if "vit" in arch else b"bn1.weight",
                                          buf.ctypes.data_as(ctypes.c_void_p), 3) == -1
        assert b"expects" in L.effocr_last_error()
        assert L.effocr_encoder_upload(h, ctypes.c_void_p(16), 1 << 40) == -5      # parameters never set
        assert L.effocr_encoder_forward(h, ctypes.c_void_p(16), 1, ctypes.c_void_p(16), 0, ctypes.c_void_p(16), 1 << 40, None) == -5
    finally:
        L.effocr_encoder_destroy(h)


def test_create_rejects_bad_requests(hip_lib):
    h = ctypes.c_void_p()
    assert hip_lib.effocr_encoder_create(b"xcit_small_12_p8_224", 224, 0, ctypes.byref(h)) == -2
    assert b"unsupported architecture" in hip_lib.effocr_last_error()
    assert hip_lib.effocr_encoder_create(b"vit_small_patch16_224", 100, 0, ctypes.byref(h)) == -1
    assert hip_lib.effocr_encoder_create(b"vit_small_patch16_224", 160, 0, ctypes.byref(h)) == -2    # 101 tokens: no attention kernel
    assert hip_lib.effocr_encoder_create(b"vit_small_patch16_224", 224, 7, ctypes.byref(h)) == -1
    assert hip_lib.effocr_encoder_create(None, 224, 0, ctypes.byref(h)) == -1


def test_knn_workspace_and_argument_checks(hip_lib):
    L = hip_lib
    assert L.effocr_knn_workspace_bytes(1024, 10000, 384, 10) >= 40 * 1024 * 16 * 8
    assert L.effocr_knn_workspace_bytes(64, 96, 512, 10) <= 512                 # single chunk: no partial lists
    assert L.effocr_knn_workspace_bytes(0, 100, 64, 1) == 0
    p = ctypes.c_void_p(16)
    # k > 32: the exact search runs in passes of 32 columns (GPU tests); the screened entry point refuses and names the exact one
    assert L.effocr_knn_ip_topk_screened(p, 4, p, p, 100000, 384, 33, ctypes.c_float(1.0), p, p, p, 1 << 40, None) == -2
    assert b"exact multi-pass" in L.effocr_last_error()
    assert L.effocr_knn_ip_topk(p, 4, p, 10, 100, 5, p, p, p, 1 << 30, None) == -2      # d % 32 != 0
    assert L.effocr_knn_ip_topk(p, 4, p, 10, 384, 0, p, p, p, 1 << 30, None) == -1
    assert L.effocr_knn_ip_topk(p, 1024, p, 10000, 384, 10, p, p, p, 16, None) == -3    # workspace too small
    assert L.effocr_knn_ip_topk(None, 4, p, 10, 384, 5, p, p, p, 1 << 30, None) == -1
    assert L.effocr_knn_ip_topk(p, 0, p, 10, 384, 5, p, p, p, 0, None) == 0            # empty query batch is a no-op


def test_fast_path_operator_argument_checks(hip_lib):
    """The blocked-layout operators and the crop transform reject bad requests before any device work
    (dummy non-NULL pointers are never dereferenced on these paths)."""
    L = hip_lib
    p = ctypes.c_void_p(4096)
    f3 = (ctypes.c_float * 3)(1.0, 1.0, 1.0)
    z3 = (ctypes.c_float * 3)(0.0, 1.0, 1.0)
    # crop transform: geometry, NULLs, std == 0, output size, box count
    assert L.effocr_crop_transform(p, 0, 10, 30, p, 1, 224, 1, f3, f3, f3, p, None) == -1
    assert L.effocr_crop_transform(p, 10, 10, 29, p, 1, 224, 1, f3, f3, f3, p, None) == -1            # row stride < 3*width
    assert L.effocr_crop_transform(None, 10, 10, 30, p, 1, 224, 1, f3, f3, f3, p, None) == -1
    assert L.effocr_crop_transform(p, 10, 10, 30, p, 1, 224, 1, f3, z3, f3, p, None) == -1
    assert L.effocr_crop_transform(p, 10, 10, 30, p, 1, 30, 1, f3, f3, f3, p, None) == -2             # size % 4
    # the batch form: image count / strides, NULLs, std == 0, size % 4
    assert L.effocr_crop_transform_batch(p, 0, 300, 10, 10, 30, p, 1, 224, 1, f3, f3, f3, p, None) == -1
    assert L.effocr_crop_transform_batch(p, 2, 299, 10, 10, 30, p, 1, 224, 1, f3, f3, f3, p, None) == -1     # image stride < rows * row stride
    assert L.effocr_crop_transform_batch(p, 2, 300, 10, 10, 30, None, 1, 224, 1, f3, f3, f3, p, None) == -1
    assert L.effocr_crop_transform_batch(p, 2, 300, 10, 10, 30, p, 1, 224, 1, f3, z3, f3, p, None) == -1
    assert L.effocr_crop_transform_batch(p, 2, 300, 10, 10, 30, p, 1, 30, 1, f3, f3, f3, p, None) == -2
    assert L.effocr_crop_transform_batch(p, 2, 300, 10, 10, 30, p, 0, 224, 1, f3, f3, f3, p, None) == 0
    assert L.effocr_crop_transform(p, 10, 10, 30, p, 0, 224, 1, f3, f3, f3, p, None) == 0
    # gemm3 / blocked LayerNorm
    assert L.effocr_op_linear_blocked(0, 0, p, p, p, None, p, 32, 128, 128, 32, None) == -2            # N % 192 / 256
    assert L.effocr_op_linear_blocked(0, 2, p, p, p, None, p, 32, 192, 128, 32, None) == -1            # residual epilogue without residual
    assert L.effocr_op_linear_blocked(0, 5, p, p, p, None, p, 32, 192, 128, 32, None) == -1
    assert L.effocr_op_layernorm_blocked(2, p, 32, 384, p, p, 1e-6, p, None) == -2                     # fp32 output
    assert L.effocr_op_layernorm_blocked(0, p, 32, 100, p, p, 1e-6, p, None) == -2
    # fused MLP (+ projection), row-block linears
    assert L.effocr_op_mlp_blocked(0, p, p, p, 1e-6, p, p, p, p, p, 32, 256, 1024, 32, None, 0, None) == -2
    assert L.effocr_op_mlp_blocked(2, p, p, p, 1e-6, p, p, p, p, p, 32, 384, 1536, 32, None, 0, None) == -2   # fp32
    assert L.effocr_op_mlp_blocked(0, p, p, p, 1e-6, p, p, p, p, p, 33, 384, 1536, 32, None, 0, None) == -1   # rows_alloc < m
    assert L.effocr_op_mlp_blocked(0, None, p, p, 1e-6, p, p, p, p, p, 32, 384, 1536, 32, None, 0, None) == -1
    assert L.effocr_op_mlp_blocked(0, p, p, p, 1e-6, p, p, p, p, p, 0, 384, 1536, 0, None, 0, None) == 0
    assert L.effocr_op_proj_mlp_blocked(0, p, None, p, p, p, p, 1e-6, p, p, p, p, p, 32, 384, 1536, 32, None, 0, None) == -1


def test_screened_knn_argument_checks(hip_lib):
    L = hip_lib
    p = ctypes.c_void_p(4096)
    assert L.effocr_knn_screen_workspace_bytes(1024, 1_000_000, 768, 10) >= 1024 * 512 * 4 + 1024 * 768 * 2
    assert L.effocr_knn_screen_workspace_bytes(0, 100, 64, 1) == 0
    call = lambda nq, n, d, k, ws: L.effocr_knn_ip_topk_screened(p, nq, p, p, n, d, k, 1.0, p, p, p, ws, None)
    assert call(4, 1000, 96, 5, 1 << 30) == -2          # d % 64
    assert call(4, 3, 128, 5, 1 << 30) == -2            # fewer rows than k
    assert call(4, 1000, 128, 33, 1 << 30) == -2        # k > 32
    assert call(4, 1000, 128, 5, 16) == -3              # workspace too small
    assert call(0, 1000, 128, 5, 0) == 0
    assert L.effocr_knn_ip_topk_screened(p, 4, p, None, 1000, 128, 5, 1.0, p, p, p, 1 << 30, None) == -1
    assert L.effocr_knn_ip_topk_screened(p, 4, p, p, 1000, 128, 5, float("nan"), p, p, p, 1 << 30, None) == -1
    assert L.effocr_convert_bf16(p, -1, p, None) == -1
    assert L.effocr_convert_bf16(None, 8, p, None) == -1
    assert L.effocr_convert_bf16(p, 0, p, None) == 0
