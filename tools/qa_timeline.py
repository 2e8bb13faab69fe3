#!/usr/bin/env python
"""Per-head timeline of the fused qkv + attention kernel from a -DQA_STAMP build (tools/ab_build.sh qastamp "-DQA_STAMP" qkvattn.hip):
   EFFOCR_HIP_LIB=$PWD/tools/ab/lib_qastamp.so python tools/qa_timeline.py [batch]
Every wave stamps s_memtime at the milestones of a head (the last head it ran stays in the table); prints the mean length of each segment
per wave index (waves 0-2 own two full token tiles, wave 3 owns tile 6 + the dummy) and the whole head."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effocr_amd.encoders import HipEncoder
from effocr_amd.weights import init_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
enc = HipEncoder("vit_small_patch16_224", init_state_dict("vit_small_patch16_224", seed=0, img_size=224), img_size=224, precision="bf16", device=dev)
enc.set_option("cls_only_last", 0)              # every block runs the all-token kernel: the last launch is a full one
x = torch.randn(B, 3, 224, 224, device=dev)
for _ in range(3):
    enc.forward(x, normalize=True)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["EFFOCR_HIP_LIB"])
NW, NS = 256 * 4, 16
buf = (ctypes.c_ulonglong * (NW * NS))()
assert lib.effocr_debug_qa_stamps(buf, NW * NS) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4, NS).astype(np.int64)
names = [(0, 1, "q projection + pack"), (1, 2, "k projection + pack + LDS"), (2, 3, "v projection + pack + LDS"), (3, 4, "K/V barrier"),
         (4, 5, "tile 0: Q K^T"), (5, 6, "tile 0: row max"), (6, 7, "tile 0: exp + P V"), (7, 8, "tile 0: normalise + store"),
         (8, 10, "tile 1: Q K^T"), (10, 11, "tile 1: row max"), (11, 12, "tile 1: exp + P V"), (12, 13, "tile 1: normalise + store")]
for w in range(4):
    tw = t[:, w]
    tw = tw[tw[:, 15] > tw[:, 0]]
    tot = tw[:, 15] - tw[:, 0]
    print(f"wave {w}: {len(tw)} workgroups, ticks per head mean {tot.mean():.0f} (p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}); "
          f"kernel entry -> start of the LAST head the workgroup ran (one head per workgroup in a small call: = the prologue) {(tw[:, 0] - tw[:, 14]).mean():.0f}")
    for a, b, n in names:
        if w == 3 and a >= 8:
            continue
        seg = tw[:, b] - tw[:, a]
        print(f"    {n:30s} {seg.mean():8.0f}  {100 * seg.mean() / tot.mean():5.1f} %")
