#!/usr/bin/env python
"""Crop pre-processing kernel alone at BASELINE configs[4]'s shape: 16 line images of 4096 x 256, ~71 character boxes per line (1 139 crops of
roughly 40 x 60 pixels -> [n, 3, 224, 224] fp32), plus down-scaling boxes (the direct fallback).  Prints the time per launch and a checksum
(sum of |x| in float64 and an xor of the raw bits) so that two libraries can be compared bit for bit:
   [EFFOCR_HIP_LIB=tools/ab/lib_<variant>.so] python tools/crop_time.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effocr_amd.transforms import PairedTransform

dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
imgs = torch.from_numpy(rng.integers(0, 256, (16, 256, 4096, 3), dtype=np.uint8)).to(dev)
rows = []
for li in range(16):
    x = 10
    for _ in range(71):
        w, h = int(rng.integers(24, 64)), int(rng.integers(40, 200))
        y0 = int(rng.integers(0, 256 - h))
        rows.append((x, y0, x + w, y0 + h, li))
        x += w + int(rng.integers(2, 8))
rows.append((0, 0, 900, 256, 3)); rows.append((100, 0, 700, 250, 5)); rows.append((5, 5, 5, 50, 1))      # down-scaling crops, an empty box
bx = torch.tensor(rows, dtype=torch.int32, device=dev)
# run_effocr's own boxes (infer_effocr_onnx_multi.py:313-320): double-clipped to the full line height — 256-tall strips, padded to 256 x 256
bx_c5 = bx.clone(); bx_c5[:, 1] = 0; bx_c5[:, 3] = 256
for aa, bx in ((True, bx), (False, bx), (True, bx_c5)):
    tf = PairedTransform(size=224, antialias=aa, device=dev)
    out = tf.boxes_batch(imgs, bx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        tf.boxes_batch(imgs, bx, out=out)
    e1.record()
    torch.cuda.synchronize()
    bits = out.view(torch.int32)
    x = int(torch.bitwise_xor(bits.flatten()[0::2], bits.flatten()[1::2]).sum(dtype=torch.int64))
    print(f"{os.environ.get('EFFOCR_HIP_LIB', 'product library')}: antialias={aa} {len(rows)} crops {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch, "
          f"{out.numel() * 4 / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12:.2f} TB/s of stores; sum|x| {out.double().abs().sum().item():.6f} bits {x}")
