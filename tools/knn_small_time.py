#!/usr/bin/env python
"""Exact search at the reference's call sizes against a 10 000-row index: tile kernel vs streaming kernel (knn_set_option stream_min_rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from effocr_amd import _lib
from effocr_amd.knn import IndexFlatIP
dev = torch.device("cuda:0")
L = _lib.lib()
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for N in (10000, 30000):
    g = torch.Generator(device=dev).manual_seed(1)
    X = torch.nn.functional.normalize(torch.randn(N, 384, generator=g, device=dev), dim=1)
    idx = IndexFlatIP(384, device=dev, screen=False); idx.add(X)
    for B in (33, 64, 100, 128, 200, 256):
        Q = torch.nn.functional.normalize(torch.randn(B, 384, generator=g, device=dev), dim=1)
        for k in (10, 1):
            row = []
            ref = None
            for name, v in (("tile", 65536), ("stream", 4096)):
                _lib.check(L.effocr_knn_set_option(b"stream_min_rows", v), "opt")
                d, i = idx.search_device(Q, k)
                if ref is None: ref = (d.clone(), i.clone())
                else: assert torch.equal(i, ref[1]) and torch.equal(d.view(torch.int32), ref[0].view(torch.int32))
                row.append(f"{name} {t(lambda: idx.search_device(Q, k)):7.1f} us")
            print(f"N={N} B={B} k={k}: " + "  ".join(row), flush=True)
_lib.check(L.effocr_knn_set_option(b"stream_min_rows", 65536), "opt")
