#!/usr/bin/env python
"""ViT-B/16: HipEncoder.forward with the call cut into 1..3 concurrent sub-batches (split_streams forced), 3 interleaved rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from effocr_amd.encoders import HipEncoder
from effocr_amd.weights import init_state_dict
dev = torch.device("cuda:0")
arch = "vit_base_patch16_224"
enc = HipEncoder(arch, init_state_dict(arch, seed=0, img_size=224), precision="bf16", device=dev)
for B in (256, 512, 1024):
    x = torch.randn(B, 3, 224, 224, device=dev)
    for rnd in range(3):
        row = []
        for S in (1, 2, 3):
            enc.split_streams = S
            step = lambda: enc.forward(x, normalize=True)
            for _ in range(3): step()
            torch.cuda.synchronize(); n = max(6, 4096 // B); t0 = time.perf_counter()
            for _ in range(n): step()
            torch.cuda.synchronize(); row.append(B * n / (time.perf_counter() - t0))
        print(f"B={B:5d} round {rnd}: " + "  ".join(f"{S} part{'s' if S > 1 else ' '} {r:8.0f}" for S, r in zip((1, 2, 3), row)), flush=True)
