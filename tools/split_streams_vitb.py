#!/usr/bin/env python
"""ViT-B/16 (configs[3]'s encoder): a 1024-crop call as S concurrent sub-calls on S streams — does an HBM-bound attention beside an
MFMA-bound GEMM of the other sub-batch pay?  python tools/split_streams_vitb.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from effocr_amd.encoders import HipEncoder
from effocr_amd.weights import init_state_dict
dev = torch.device("cuda:0")
arch = "vit_base_patch16_224"
enc = HipEncoder(arch, init_state_dict(arch, seed=0, img_size=224), precision="bf16", device=dev)
for T in (512, 1024):
    row = []
    for S in (1, 2, 3, 4):
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        xs = [torch.randn(T // S, 3, 224, 224, device=dev) for _ in range(S)]
        calls = 8
        def run(i, n):
            with torch.cuda.stream(streams[i]):
                for _ in range(n):
                    enc.forward(xs[i], normalize=True)
                streams[i].synchronize()
        for i in range(S): run(i, 2)
        torch.cuda.synchronize()
        ths = [threading.Thread(target=run, args=(i, calls)) for i in range(S)]
        t0 = time.perf_counter()
        for th in ths: th.start()
        for th in ths: th.join()
        torch.cuda.synchronize()
        row.append(f"{(T // S) * S * calls / (time.perf_counter() - t0):8.0f}")
    print(f"T={T:5d}: " + "  ".join(f"{S} stream{'s' if S > 1 else ' '} {r}" for S, r in zip((1, 2, 3, 4), row)) + "  crops/s", flush=True)
