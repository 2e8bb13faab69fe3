"""Streaming k-NN (<= 64 queries): fixed cost vs streaming rate (GPU box).  Times the exact search at several index sizes and fits
t = F + bytes / R per (D, B), so the launch / prologue / merge overhead F is separated from the HBM rate R the loop sustains.
    python tools/knn_fixed.py [--dims 384,768] [--batches 1,16,64] [--rows 125000,250000,500000,1000000] [--opt NAME=INT ...]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effocr_amd import _lib                      # noqa: E402
from effocr_amd.knn import IndexFlatIP           # noqa: E402


def tgpu(fn, dev, iters=30, warm=5):
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
    ap.add_argument("--rows", default="125000,250000,500000,1000000")
    ap.add_argument("--dims", default="384,768")
    ap.add_argument("--batches", default="1,16,64")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    L = _lib.lib()
    for o in a.opt:
        name, val = o.split("=")
        _lib.check(L.effocr_knn_set_option(name.encode(), int(val)), "knn_set_option")
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    rows = [int(v) for v in a.rows.split(",")]
    for D in [int(v) for v in a.dims.split(",")]:
        xb_all = torch.nn.functional.normalize(torch.randn(max(rows), D, generator=g, device=dev), dim=1)
        for B in [int(v) for v in a.batches.split(",")]:
            ts = []
            for N in rows:
                ex = IndexFlatIP(D, device=dev, screen=False)
                ex.add(xb_all[:N])
                q = torch.nn.functional.normalize(xb_all[:B] + 0.1 * torch.randn(B, D, generator=g, device=dev), dim=1)
                ts.append(tgpu(lambda: ex.search_device(q, a.k), dev))
                del ex
            by = np.array([N * D * 4.0 for N in rows])
            A = np.stack([np.ones_like(by), by], 1)
            (F, invR), *_ = np.linalg.lstsq(A, np.array(ts), rcond=None)
            line = "  ".join(f"N={N // 1000}k {1e6 * t:7.1f} us ({N * D * 4 / t / 8e12:.3f})" for N, t in zip(rows, ts))
            print(f"D={D} B={B:3d} k={a.k}: {line}   fit: F = {1e6 * F:6.1f} us, R = {1e-12 / invR:5.2f} TB/s", flush=True)
        del xb_all
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
