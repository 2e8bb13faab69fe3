"""k-NN timing sweep (GPU box): exact vs screened search, queries x rows x dim; HBM / fp32-MFMA roof fractions.
    python tools/knn_sweep.py [--rows 10000,1000000] [--dims 384,768] [--batches 1,16,64,128,1024]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effocr_amd.knn import IndexFlatIP   # noqa: E402


def tgpu(fn, dev, iters=20, warm=3):
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
    ap.add_argument("--rows", default="10000,1000000")
    ap.add_argument("--dims", default="384,768")
    ap.add_argument("--batches", default="1,16,32,64,128,256,1024")
    ap.add_argument("--k", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    for D in [int(v) for v in a.dims.split(",")]:
        for N in [int(v) for v in a.rows.split(",")]:
            xb = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
            ex = IndexFlatIP(D, device=dev, screen=False)
            ex.add(xb)
            sc = IndexFlatIP(D, device=dev, screen=True)
            sc._xb = ex._xb
            for B in [int(v) for v in a.batches.split(",")]:
                q = torch.nn.functional.normalize(xb[:B] + 0.1 * torch.randn(B, D, generator=g, device=dev), dim=1)
                te = tgpu(lambda: ex.search_device(q, a.k), dev)
                ts = tgpu(lambda: sc.search_device(q, a.k), dev)
                same = torch.equal(ex.search_device(q, a.k)[1], sc.search_device(q, a.k)[1])
                print(f"D={D:4d} N={N:8d} B={B:5d} k={a.k}: exact {1e3 * te:8.3f} ms (HBM {N * D * 4 / te / 8e12:6.3f}, fp32-MFMA {2.0 * B * N * D / te / 157.3e12:6.3f})"
                      f"   screened {1e3 * ts:8.3f} ms   ids equal {same}", flush=True)
            del ex, sc, xb
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
