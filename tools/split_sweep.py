#!/usr/bin/env python
"""HipEncoder.forward + k-NN per call size with the call cut into 1..4 concurrent sub-batches (split_streams forced): crops/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from effocr_amd.encoders import HipEncoder
from effocr_amd.knn import IndexFlatIP
from effocr_amd.weights import init_state_dict
dev = torch.device("cuda:0")
arch = "vit_small_patch16_224"
enc = HipEncoder(arch, init_state_dict(arch, seed=0, img_size=224), precision="bf16", device=dev)
idx = IndexFlatIP(384, device=dev)
idx.add(torch.nn.functional.normalize(torch.randn(10000, 384, generator=torch.Generator().manual_seed(0)), dim=1))
SIZES = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else (160, 192, 224, 256, 320, 384, 448, 512, 640, 768, 1024)
for B in SIZES:
    x = torch.randn(B, 3, 224, 224, device=dev)
    row = []
    for S in (1, 2, 3, 4):
        enc.split_streams = S
        step = lambda: idx.search_device(enc.forward(x, normalize=True), 10)
        for _ in range(5): step()
        torch.cuda.synchronize(); n = max(30, 8192 // B); t0 = time.perf_counter()
        for _ in range(n): step()
        torch.cuda.synchronize(); row.append(B * n / (time.perf_counter() - t0))
    best = max(range(4), key=lambda i: row[i]) + 1
    print(f"B={B:5d}: " + "  ".join(f"{S} part{'s' if S > 1 else ' '} {r:8.0f}" for S, r in zip((1, 2, 3, 4), row)) + f"   best {best}", flush=True)
