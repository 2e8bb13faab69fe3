#!/usr/bin/env python
"""Forward time of the YOLOv5s localizer network (16 images of 640 x 640, device-resident), median of N rounds of 10 forwards:
   [EFFOCR_HIP_LIB=tools/ab/lib_<variant>.so] python tools/loc_time.py [rounds]   (same-box A/B of convolution variants)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effocr_amd.localizer_engine import HipLocalizer, init_yolov5s_state_dict

dev = torch.device("cuda:0")
loc = HipLocalizer(init_yolov5s_state_dict(2, seed=0), input_shape=(640, 640), device=dev)
NB = int(os.environ.get("LOC_BATCH", "16"))
im = torch.rand(NB, 3, 640, 640, device=dev)
for _ in range(5):
    loc.forward(im)
torch.cuda.synchronize()
ts = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        loc.forward(im)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
print(f"{os.environ.get('EFFOCR_HIP_LIB', 'product library')}: {NB} images, forward {np.median(ts):.3f} ms (min {min(ts):.3f}), {np.median(ts) / NB * 1e3:.1f} us per image")
