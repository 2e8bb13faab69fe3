#!/usr/bin/env python
"""Generates the committed golden vectors under tests/golden/ (run in the build container, CPU only).

The reference cannot be imported offline (timm / faiss / pytorch_metric_learning / onnxruntime are
absent), so the vectors are produced by the repo's own CPU restatement AFTER it has been
cross-checked against the independent ``transformers`` implementations (oracle/hf_crosscheck.py):
every encoder vector below is asserted to agree between oracle A and oracle B before it is written.
Weights are NOT stored: they are regenerated from their seed (effocr_amd.weights.init_state_dict).

Files (all small .npz):
  enc_<arch>.npz      seed, img, x [B,3,img,img] f32, emb [B,D] f32 (pooled, not normalised)
  knn_c2small.npz     Q [32,384], X [1000,384] (unit rows), k=10: D, I  (C oracle, ascending-k fmaf)
  knn_ties.npz        exact-duplicate rows: pins the "lower id wins" rule
  pipeline.npz        create_batches case (70 crops, one None) + blacklist/compaction case
  crop_transform.npz  image [48,80,3] u8, boxes [6,4] (python-slice semantics incl. negative / oversize), size 32:
                      out_aa, out_plain [6,3,32,32] f32 (numpy restatement asserted equal to torch's
                      interpolate first).  `python make_golden.py crop` regenerates only this file.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from effocr_amd.weights import init_state_dict          # noqa: E402
from oracle import knn_ref                               # noqa: E402
from oracle.encoders_ref import encoder_forward          # noqa: E402
from oracle.hf_crosscheck import hf_encoder_forward      # noqa: E402
from oracle import crop_transform_ref                    # noqa: E402


def crop_golden():
    rng = np.random.default_rng(21)
    yy, xx = np.mgrid[0:48, 0:80]
    img = 190 + 50 * np.sin(xx / 5.0)[..., None] * np.cos(yy / 3.0)[..., None] + rng.normal(0, 12, (48, 80, 3))
    img[10:40, 20:24] = 15
    img[22:26, 8:70] = 40
    img = np.clip(img, 0, 255).astype(np.uint8)
    boxes = np.array([[2, 3, 30, 45], [0, 0, 80, 48], [50, 10, 57, 40], [5, 20, 75, 27], [-20, 4, 80, 44], [60, -8, 200, 100]], np.int32)
    outs = {}
    for aa in (True, False):
        a = crop_transform_ref.transform_boxes(img, boxes, size=32, antialias=aa, use_torch=True)
        b = crop_transform_ref.transform_boxes(img, boxes, size=32, antialias=aa, use_torch=False)
        assert np.abs(a - b).max() <= (1e-5 if aa else 2e-4), (aa, np.abs(a - b).max())
        outs["out_aa" if aa else "out_plain"] = a
    np.savez_compressed(os.path.join(HERE, "crop_transform.npz"), image=img, boxes=boxes, size=32, **outs)
    print("crop_transform", outs["out_aa"].shape)


def unit(a):
    return (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)


def main():
    crop_golden()
    if sys.argv[1:] == ["crop"]:
        return
    for arch, img, B, seed in [("resnet18", 32, 8, 11), ("vit_tiny_test", 64, 4, 12),
                               ("vit_small_patch16_224", 224, 4, 13), ("vit_base_patch16_224", 224, 2, 14)]:
        sd = init_state_dict(arch, seed=seed, img_size=img)
        x = torch.randn(B, 3, img, img, generator=torch.Generator().manual_seed(seed + 1000))
        a = encoder_forward(arch, sd, x)
        b = hf_encoder_forward(arch, sd, x)
        assert ((a - b).abs().max() / a.abs().max()).item() < 1e-5, arch
        np.savez_compressed(os.path.join(HERE, f"enc_{arch}.npz"), seed=seed, img=img,
                            x=x.numpy().astype(np.float16 if arch.startswith("vit_") and img == 224 else np.float32),
                            emb=a.numpy())
        # 224x224 inputs are stored as float16 to keep the fixture small; the test feeds x.astype(float32)
        if arch.startswith("vit_") and img == 224:
            x16 = torch.from_numpy(x.numpy().astype(np.float16).astype(np.float32))
            a = encoder_forward(arch, sd, x16)
            b = hf_encoder_forward(arch, sd, x16)
            assert ((a - b).abs().max() / a.abs().max()).item() < 1e-5, arch
            np.savez_compressed(os.path.join(HERE, f"enc_{arch}.npz"), seed=seed, img=img,
                                x=x.numpy().astype(np.float16), emb=a.numpy())
        print(arch, "emb", tuple(a.shape))

    rng = np.random.default_rng(2024)
    X = unit(rng.standard_normal((1000, 384)))
    Q = unit(X[rng.permutation(1000)[:32]] + 0.1 * rng.standard_normal((32, 384)).astype(np.float32))
    D, I = knn_ref.flat_ip_search(Q, X, 10)
    D64, I64 = knn_ref.flat_ip_search_f64(Q, X, 10)
    assert np.array_equal(I, I64) and np.abs(D - D64).max() < 1e-6
    np.savez_compressed(os.path.join(HERE, "knn_c2small.npz"), Q=Q, X=X, k=10, D=D, I=I)

    base = unit(rng.standard_normal((40, 128)))
    Xt = np.concatenate([base, base[:20], base[5:9], base])
    Qt = base[:16].copy()
    Dt, It = knn_ref.flat_ip_search(Qt, Xt, 10)
    _, It64 = knn_ref.flat_ip_search_f64(Qt, Xt, 10)
    assert np.array_equal(It, It64)          # duplicates are bit-identical rows -> exact ties in fp64 too
    np.savez_compressed(os.path.join(HERE, "knn_ties.npz"), Q=Qt, X=Xt, k=10, D=Dt, I=It)

    # host-logic cases (infer_effocr_onnx_multi.py:143-158; infer_effocr.py:209-212)
    chars = [chr(0x3041 + i) for i in range(12)]
    blacklist = chars[3] + chars[7]
    Xb = unit(rng.standard_normal((12, 128)))
    keep = [i for i in range(12) if chars[i] not in blacklist]
    Qb = Xb[[3, 7, 0, 11]]
    Db, Ib = knn_ref.flat_ip_search(Qb, Xb[keep], 2)
    np.savez_compressed(os.path.join(HERE, "pipeline.npz"), chars=np.array(chars), blacklist=blacklist, X=Xb, Q=Qb,
                        I_after=Ib, kept=np.array(keep), n_crops=70, none_at=5, n_batches=2, pad_rows=58)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
