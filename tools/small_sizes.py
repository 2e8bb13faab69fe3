#!/usr/bin/env python
"""Per-call latency of the recognizer step (encoder + fused L2 normalise + k-NN, k = 10, 10 000-row index) at the reference's real call
sizes: one text line = B characters (infer_effocr.py:313-319), 64-crop batches (infer_effocr_onnx_multi.py:157), 128 per rank (configs[2]).
   python tools/small_sizes.py [--precision bf16] [opt=val ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from effocr_amd.encoders import HipEncoder
from effocr_amd.knn import IndexFlatIP
from effocr_amd.weights import init_state_dict
prec = sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "bf16"
dev = torch.device("cuda:0")
arch = "vit_small_patch16_224"
enc = HipEncoder(arch, init_state_dict(arch, seed=0, img_size=224), precision=prec, device=dev)
for o in sys.argv[1:]:
    if "=" in o and not o.startswith("--"):
        k, v = o.split("="); enc.set_option(k, int(v))
idx = IndexFlatIP(384, device=dev)
idx.add(torch.nn.functional.normalize(torch.randn(10000, 384, generator=torch.Generator().manual_seed(0)), dim=1))
full = None
for B in (1024, 512, 256, 128, 64, 32, 16, 8, 1):
    x = torch.randn(B, 3, 224, 224, device=dev)
    def step():
        return idx.search_device(enc.forward(x, normalize=True), 10)
    for _ in range(5): step()
    torch.cuda.synchronize()
    n = max(30, 8192 // B)
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n
    enc.profile_begin(); step(); tab = enc.profile_collect()
    # k-NN alone
    emb = enc.forward(x, normalize=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): idx.search_device(emb, 10)
    torch.cuda.synchronize(); tk = (time.perf_counter() - t0) / 50
    full = full or B / t
    print(f"B={B:5d}  {1e3 * t:8.3f} ms/call  {B / t:9.1f} crops/s  frac {B / t / full:6.3f}  encoder launches {sum(v['launches'] for v in tab.values()):3d}  "
          f"kernel ms {sum(v['ms'] for v in tab.values()):7.3f}  knn {1e6 * tk:6.1f} us", flush=True)
    if "--table" in sys.argv:
        for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["ms"]):
            print(f"        {k:22s} x{v['launches']:3d} {1e3 * v['ms']:8.1f} us")
