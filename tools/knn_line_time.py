#!/usr/bin/env python
"""Exact search at per-line call sizes (1..32 queries, k = 10) against a 10 000-row index: chunk count (wg_target) and kernel choice."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from effocr_amd import _lib
from effocr_amd.knn import IndexFlatIP
dev = torch.device("cuda:0")
L = _lib.lib()
def t(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
N = 10000
g = torch.Generator(device=dev).manual_seed(1)
X = torch.nn.functional.normalize(torch.randn(N, 384, generator=g, device=dev), dim=1)
idx = IndexFlatIP(384, device=dev, screen=False); idx.add(X)
for B in (1, 8, 16, 32, 64):
    Q = torch.nn.functional.normalize(torch.randn(B, 384, generator=g, device=dev), dim=1)
    row = []
    for force in (0, 1):
        _lib.check(L.effocr_knn_set_option(b"force_tile", force), "opt")
        for wg in (32, 64, 128, 256, 512):
            _lib.check(L.effocr_knn_set_option(b"wg_target", wg), "opt")
            idx._ws = {}
            row.append(f"{'tile' if force else 'strm'}/{wg}: {t(lambda: idx.search_device(Q, 10)):5.1f}")
    print(f"B={B:3d}  " + "  ".join(row), flush=True)
_lib.check(L.effocr_knn_set_option(b"force_tile", 0), "opt"); _lib.check(L.effocr_knn_set_option(b"wg_target", 512), "opt")
