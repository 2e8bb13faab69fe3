#!/usr/bin/env python
"""Two 64-crop sub-batches of a 128-crop call on two streams, the second one started LATE by a spin kernel so that one's fused MLP
(99 CUs) runs beside the other's per-image kernel (128 CUs): does a forced phase shift pay?  python tools/split_stagger.py [opt=val ...]"""
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
for T in (128, 64):
    for delay_us in (0, 20, 40, 60, 80, 100):
        S = 2
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        xs = [torch.randn(T // S, 3, 224, 224, device=dev) for _ in range(S)]
        calls = 40
        def run(i, n, delay):
            with torch.cuda.stream(streams[i]):
                for _ in range(n):
                    if delay and i == 1:
                        torch.cuda._sleep(int(delay * 2000))          # ~2 GHz: the SECOND sub-batch of every call starts late
                    idx.search_device(enc.forward(xs[i], normalize=True), 10)
                streams[i].synchronize()
        for i in range(S): run(i, 3, 0)
        torch.cuda.synchronize()
        ths = [threading.Thread(target=run, args=(i, calls, delay_us)) for i in range(S)]
        t0 = time.perf_counter()
        for th in ths: th.start()
        for th in ths: th.join()
        torch.cuda.synchronize()
        print(f"T={T} as 2 x {T // 2}, second sub-batch delayed {delay_us:3d} us per call: {T * calls / (time.perf_counter() - t0):8.0f} crops/s", flush=True)
