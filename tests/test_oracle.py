"""-m "not gpu": the oracle pinned against the committed golden vectors, against the independent
``transformers`` implementations (oracle B) and against a float64 restatement."""
import os

import numpy as np
import pytest
import torch

from effocr_amd.weights import init_state_dict, param_shapes
from oracle import knn_ref
from oracle.encoders_ref import encoder_forward, l2_normalize

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


@pytest.mark.parametrize("arch", ["resnet18", "vit_tiny_test", "vit_small_patch16_224", "vit_base_patch16_224"])
def test_encoder_oracle_matches_golden(arch):
    g = load(f"enc_{arch}.npz")
    sd = init_state_dict(arch, seed=int(g["seed"]), img_size=int(g["img"]))
    x = torch.from_numpy(g["x"].astype(np.float32))
    if arch == "vit_base_patch16_224":
        x = x[:1]
    emb = encoder_forward(arch, sd, x).numpy()
    ref = g["emb"][: emb.shape[0]]
    assert np.abs(emb - ref).max() <= 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("arch,img,B", [("resnet18", 32, 3), ("resnet18", 64, 2), ("vit_tiny_test", 64, 3),
                                        ("vit_small_patch16_224", 224, 1)])
def test_oracle_a_equals_independent_oracle_b(arch, img, B):
    from oracle.hf_crosscheck import hf_encoder_forward
    sd = init_state_dict(arch, seed=5, img_size=img)
    x = torch.randn(B, 3, img, img, generator=torch.Generator().manual_seed(6))
    a, b = encoder_forward(arch, sd, x), hf_encoder_forward(arch, sd, x)
    assert ((a - b).abs().max() / a.abs().max()).item() < 1e-5


def test_param_tables_have_the_published_sizes():
    def count(arch):
        return sum(int(np.prod(s)) for k, s in param_shapes(arch).items() if "running_" not in k)
    assert count("vit_small_patch16_224") == 21_665_664        # timm vit_small_patch16_224, num_classes=0
    assert count("vit_base_patch16_224") == 85_798_656
    assert count("resnet18") == 11_176_512                      # timm resnet18, num_classes=0


def test_net_prefix_is_accepted():
    sd = init_state_dict("vit_tiny_test", seed=1, img_size=64)
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(2))
    a = encoder_forward("vit_tiny_test", sd, x)
    b = encoder_forward("vit_tiny_test", {"net." + k: v for k, v in sd.items()}, x)
    assert torch.equal(a, b)


def test_knn_oracle_matches_golden_and_fp64():
    g = load("knn_c2small.npz")
    D, I = knn_ref.flat_ip_search(g["Q"], g["X"], int(g["k"]))
    assert np.array_equal(I, g["I"]) and np.array_equal(D.view(np.uint32), g["D"].view(np.uint32))
    D64, I64 = knn_ref.flat_ip_search_f64(g["Q"], g["X"], int(g["k"]))
    assert np.array_equal(I, I64) and np.abs(D - D64).max() < 1e-6
    assert (np.diff(D, axis=1) <= 0).all()                       # descending
    # scores are the ascending-k fmaf chain: reproduce one by hand in float64-of-float32 steps
    q, x = g["Q"][3], g["X"][I[3, 0]]
    acc = np.float32(0)
    for a, b in zip(q, x):
        acc = np.float32(np.float64(a) * np.float64(b) + np.float64(acc))     # exact product, one rounding
    assert acc == D[3, 0]


def test_knn_oracle_tie_rule_and_padding():
    g = load("knn_ties.npz")
    D, I = knn_ref.flat_ip_search(g["Q"], g["X"], 10)
    assert np.array_equal(I, g["I"])
    for b in range(I.shape[0]):
        top = I[b][D[b] == D[b][0]]
        assert len(top) >= 3 and list(top) == sorted(top)        # duplicates ranked by ascending id
    D, I = knn_ref.flat_ip_search(g["Q"][:2], g["X"][:3], 5)
    assert (I[:, 3:] == -1).all() and (D[:, 3:] == np.float32(knn_ref.NEG)).all()
    D, I = knn_ref.flat_ip_search(g["Q"][:2], g["X"][:0], 2)
    assert (I == -1).all()


def test_l2_normalize_oracles_agree():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((9, 384)).astype(np.float32) * 7
    x[4] = 0
    a = knn_ref.l2_normalize(x)
    b = l2_normalize(torch.from_numpy(x)).numpy()
    t = torch.nn.functional.normalize(torch.from_numpy(x), p=2, dim=1).numpy()
    np.testing.assert_allclose(a, t, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(b, t, rtol=2e-6, atol=1e-7)


def test_remove_ids_compaction_oracle():
    X = np.arange(40, dtype=np.float32).reshape(10, 4)
    Y = knn_ref.remove_ids(X, [0, 7, 3])
    assert Y.shape == (7, 4) and Y[0, 0] == 4 and Y[2, 0] == 16 and Y[-1, 0] == 36      # rows 1,2,4,5,6,8,9 survive


# ---------------------------------------------------------------- crop pre-processing
def test_crop_transform_restatement_matches_torch_and_golden():
    from oracle import crop_transform_ref as R
    g = load("crop_transform.npz")
    img, boxes, size = g["image"], g["boxes"], int(g["size"])
    for aa, key in ((True, "out_aa"), (False, "out_plain")):
        via_torch = R.transform_boxes(img, boxes, size=size, antialias=aa, use_torch=True)
        via_numpy = R.transform_boxes(img, boxes, size=size, antialias=aa, use_torch=False)
        assert np.abs(via_torch - g[key]).max() <= 1e-6          # golden = torch path of the build container
        assert np.abs(via_numpy - g[key]).max() <= (1e-5 if aa else 2e-4)


@pytest.mark.parametrize("hw", [(17, 23), (224, 224), (300, 120), (1, 1), (3, 500), (448, 448)])
def test_resize_weights_against_torch_interpolate(hw):
    from oracle import crop_transform_ref as R
    crop = np.random.default_rng(hw[0]).integers(0, 256, (*hw, 3), dtype=np.uint8)
    x = R.pad_square_to_float(crop)
    assert x.shape == (3, max(hw), max(hw)) and x[:, -1, -1].min() == 1.0 or hw[0] == hw[1]
    for aa in (True, False):
        a, b = R.resize_bilinear(x, 64, aa), R.resize_bilinear_torch(x, 64, aa)
        assert np.abs(a - b).max() <= (1e-6 if aa else 4e-6 * max(hw) + 1e-6)
    s, w, c = R.resize_weights(max(hw), 64, True)
    assert np.allclose(w.sum(1), 1.0, atol=1e-6) and (s >= 0).all() and (s + c <= max(hw)).all()


def test_python_slice_box_semantics():
    from oracle import crop_transform_ref as R
    im = np.arange(6 * 9 * 3, dtype=np.uint8).reshape(6, 9, 3)
    for box in [(1, 2, 5, 4), (-3, 0, 9, 6), (0, -2, 20, 30), (7, 1, 3, 5), (2, 2, 2, 5)]:
        x0, y0, x1, y1 = R.python_slice_box(box, 6, 9)
        assert np.array_equal(im[y0:y1, x0:x1], im[box[1]:box[3], box[0]:box[2]])
    assert R.round_box((0.5, 1.5, 2.5, 3.49)) == (0, 2, 2, 3)     # Python round: half to even
    with pytest.raises(ValueError):
        R.pad_square_to_float(im[2:2])
