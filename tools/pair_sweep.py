#!/usr/bin/env python
"""64-token wave-pair panels of the fused proj+MLP kernel (mlp_pair) against the hidden-split parts + reduction launch, per call size:
   python tools/pair_sweep.py [--precision bf16]
encoder + fused normalise, device-resident crops; also the embedding error of both against the library's fp32 mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from effocr_amd.encoders import HipEncoder
from effocr_amd.weights import init_state_dict
prec = sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "bf16"
dev = torch.device("cuda:0")
arch = "vit_small_patch16_224"
sd = init_state_dict(arch, seed=0, img_size=224)
ref = HipEncoder(arch, sd, precision="fp32", device=dev)
encs = {}
for mode in (-1, 1):
    encs[mode] = HipEncoder(arch, sd, precision=prec, device=dev)
    encs[mode].set_option("mlp_pair", mode)
def rel(a, b): return ((a - b).abs().max() / b.abs().max()).item()
for B in (8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64, 72, 80, 83):
    x = torch.randn(B, 3, 224, 224, device=dev, generator=torch.Generator(device=dev).manual_seed(B))
    r = ref.forward(x, normalize=True)
    out = {}
    for rnd in range(2):
        for mode in (-1, 1):
            enc = encs[mode]
            for _ in range(5): y = enc.forward(x, normalize=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): y = enc.forward(x, normalize=True)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
            out.setdefault(mode, []).append(dt * 1e3)
            out[(mode, "err")] = rel(y, r)
    print(f"B={B:3d}  split parts {min(out[-1]):.3f} ms (err {out[(-1, 'err')]:.2e})   pair panels {min(out[1]):.3f} ms (err {out[(1, 'err')]:.2e})   ratio {min(out[1]) / min(out[-1]):.3f}", flush=True)
