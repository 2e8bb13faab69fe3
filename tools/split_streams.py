#!/usr/bin/env python
"""A call of T crops as S concurrent sub-calls of T/S crops on S HIP streams (one Python thread each): aggregate crops/s.
Would an internal split of 128..512-crop calls beat the single-stream kernel selection?  python tools/split_streams.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from effocr_amd.encoders import HipEncoder
from effocr_amd.knn import IndexFlatIP
from effocr_amd.weights import init_state_dict
dev = torch.device("cuda:0")
arch = "vit_small_patch16_224"
enc = HipEncoder(arch, init_state_dict(arch, seed=0, img_size=224), precision="bf16", device=dev)
enc.split_streams = False
for o in sys.argv[1:]:
    if "=" in o:
        k, v = o.split("="); enc.set_option(k, int(v)); print("option", k, v)
idx = IndexFlatIP(384, device=dev)
idx.add(torch.nn.functional.normalize(torch.randn(10000, 384, generator=torch.Generator().manual_seed(0)), dim=1))
for T in (64, 128, 192, 256):
    row = []
    for S in (1, 2, 4):
        if T // S < 32: row.append("   -   "); continue
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        xs = [torch.randn(T // S, 3, 224, 224, device=dev) for _ in range(S)]
        calls = max(20, 4096 // T)
        def run(i, n):
            with torch.cuda.stream(streams[i]):
                for _ in range(n):
                    idx.search_device(enc.forward(xs[i], normalize=True), 10)
                streams[i].synchronize()
        for i in range(S): run(i, 3)
        torch.cuda.synchronize()
        ths = [threading.Thread(target=run, args=(i, calls)) for i in range(S)]
        t0 = time.perf_counter()
        for th in ths: th.start()
        for th in ths: th.join()
        torch.cuda.synchronize()
        row.append(f"{T * calls / (time.perf_counter() - t0):8.0f}")
    print(f"T={T:5d}: 1 stream {row[0]}  2 streams {row[1]}  4 streams {row[2]}  crops/s", flush=True)
