#!/usr/bin/env python
"""Per-stage timeline of the implicit-GEMM convolution from a -DCONV_STAMP build (tools/ab_build.sh cstamp "-DCONV_STAMP" resnet.hip):
   EFFOCR_HIP_LIB=$PWD/tools/ab/lib_cstamp.so python tools/conv_timeline.py
Runs YOLOv5s forwards (16 images of 640 x 640) and prints, for the LAST 3x3 128 -> 128 layer on the 128-channel tile (200 workgroups, one per CU),
the s_memtime ticks per K-stage split into: issue of the next stage's global loads, the stage's MFMAs, wait for the loads + LDS stores,
barrier; plus prologue / epilogue and the spread over workgroups."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effocr_amd.localizer_engine import HipLocalizer, init_yolov5s_state_dict

dev = torch.device("cuda:0")
loc = HipLocalizer(init_yolov5s_state_dict(2, seed=0), input_shape=(640, 640), device=dev)
im = torch.rand(16, 3, 640, 640, device=dev)
for _ in range(4):
    loc.forward(im)
    torch.cuda.synchronize()
    time.sleep(0.02)
lib = ctypes.CDLL(os.environ["EFFOCR_HIP_LIB"])
WGS, NS = 256, 160
buf = (ctypes.c_ulonglong * (WGS * 4 * NS))()
assert lib.effocr_debug_conv_stamps(buf, WGS * 4 * NS) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(WGS, 4, NS).astype(np.int64)
t0 = t[:, :, 0]
live = t0[:, 0] > t0.max() - 2_000_000                 # workgroups of the last stamped launch
t = t[live]
print(f"{t.shape[0]} workgroups of the last stamped launch")
nst = int(((t[0, 0, 4:] > 0).sum()) // 4)
tot = t[:, :, 2] - t[:, :, 0]
print(f"stages {nst}; ticks per wave, kernel entry -> end: mean {tot.mean():.0f} (min {tot.min()}, max {tot.max()})")
print(f"  prologue (entry -> first barrier passed)    {(t[:, :, 1] - t[:, :, 0]).mean():9.0f}")
st = t[:, :, 4:4 + 4 * nst].reshape(t.shape[0], 4, nst, 4)
prev = np.concatenate([t[:, :, 1:2], st[:, :, :-1, 3]], axis=2)     # loop-top time of each stage
seg = {"issue next stage's loads": st[:, :, :, 0] - prev, "MFMAs of the stage": st[:, :, :, 1] - st[:, :, :, 0],
       "wait loads + LDS stores": st[:, :, :, 2] - st[:, :, :, 1], "barrier": st[:, :, :, 3] - st[:, :, :, 2]}
per_stage = (st[:, :, :, 3] - prev)
print(f"  per stage                                    {per_stage.mean():9.0f}   (p10 {np.percentile(per_stage, 10):.0f}, p90 {np.percentile(per_stage, 90):.0f}, max {per_stage.max()})")
for k, v in seg.items():
    print(f"    {k:34s}         {v.mean():9.0f}   {100 * v.mean() / per_stage.mean():5.1f} %  (p10 {np.percentile(v, 10):.0f}, p90 {np.percentile(v, 90):.0f}, max {v.max()})")
print(f"  epilogue (last barrier -> end)               {(t[:, :, 2] - st[:, :, -1, 3]).mean():9.0f}")
byst = per_stage.mean(axis=(0, 1))
print("  mean ticks by stage index:", " ".join(f"{v:.0f}" for v in byst))
s0, e0 = t[:, :, 0].min(), t[:, :, 2].max()
print(f"launch span (first entry -> last end) {e0 - s0} ticks; entries spread {t[:, :, 0].max() - s0}; ends spread {e0 - t[:, :, 2].min()}")
