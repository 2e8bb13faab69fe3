"""Encoder (+ k-NN) time per call vs crops per call, default dispatch vs forced variants (GPU box; A/B within one process).

    python tools/sweep_batch.py [--opts use_qkvattn=2 ...] [--batches 64,128,...] [--breakdown 64,128]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effocr_amd.encoders import HipEncoder          # noqa: E402
from effocr_amd.knn import FaissKNN, IndexFlatIP     # noqa: E402
from effocr_amd.weights import init_state_dict       # noqa: E402


def tgpu(fn, dev, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="vit_small_patch16_224")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--batches", default="32,64,96,128,192,256,512,1024")
    ap.add_argument("--variants", default="default", help="';'-separated option sets, each 'name=v,name=v' or 'default'")
    ap.add_argument("--breakdown", default="", help="batches for which the per-kernel-class table is printed")
    ap.add_argument("--knn", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sd = init_state_dict(a.arch, seed=0, img_size=224)
    knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False, device=dev)
    D = 384 if "small" in a.arch else 768
    knn.train(torch.nn.functional.normalize(torch.randn(10000, D, generator=torch.Generator().manual_seed(0)), dim=1))
    bd = {int(b) for b in a.breakdown.split(",") if b}
    for var in a.variants.split(";"):
        enc = HipEncoder(a.arch, sd, img_size=224, precision=a.precision, device=dev)
        if var != "default":
            for kv in var.split(","):
                n, _, v = kv.partition("=")
                enc.set_option(n, int(v))
        for B in [int(b) for b in a.batches.split(",")]:
            x = torch.randn(B, 3, 224, 224, device=dev)
            step = (lambda: knn(enc.forward(x, normalize=True), k=10)) if a.knn else (lambda: enc.forward(x, normalize=True))
            t = tgpu(step, dev, max(5, min(40, 4096 // B)))
            print(f"[{var}] B={B:5d}  {1e3 * t:8.3f} ms  {B / t:10.1f} crops/s", flush=True)
            if B in bd:
                enc.profile_begin()
                step()
                tab = enc.profile_collect()
                for n, v in sorted(tab.items(), key=lambda kv: -kv[1]["ms"]):
                    print(f"      {n:20s} {v['ms']:8.3f} ms x{v['launches']:3d}", flush=True)
        del enc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
