#!/usr/bin/env python
"""k-NN call time at BASELINE configs[1]'s own shapes (10 000 x 384 index) and configs[3]'s (1M x 768): exact / round-4 screen / Q-stationary screen."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from effocr_amd.knn import IndexFlatIP
dev = torch.device("cuda:0")

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

SHAPES = ((10000, 384, (64, 128, 256, 1024, 1139)), (1000000, 384, (256, 1024)), (1000000, 768, (256, 1024)))
if os.environ.get("KNN_SHAPES") == "c2":
    SHAPES = ((10000, 384, (1024,)),)
if os.environ.get("KNN_SHAPES") == "c4":
    SHAPES = ((1000000, 768, (1024,)),)
for N, D, Bs in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
    for B in Bs:
        for k in ((10,) if os.environ.get("KNN_SHAPES") else (10, 1)):
            Q = torch.nn.functional.normalize(X[torch.randint(0, N, (B,), generator=g, device=dev)] + 0.1 * torch.randn(B, D, generator=g, device=dev), dim=1)
            row = []
            for name, kw in (("exact", dict(screen=False)), ("screen_r4", dict(screen=True)), ("screen_qs", dict(screen=True))):
                idx = IndexFlatIP(D, device=dev, **kw)
                idx.use_qs = name == "screen_qs"
                idx.add(X)
                row.append(f"{name} {t(lambda: idx.search_device(Q, k), 10 if N > 100000 else 30):8.1f} us")
                del idx
            print(f"N={N} D={D} B={B} k={k}: " + "  ".join(row), flush=True)
