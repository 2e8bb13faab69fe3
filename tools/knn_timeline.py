#!/usr/bin/env python
"""Phase timeline of the streaming k-NN kernel from a -DKNN_STAMP build (tools/ab_build.sh kstamp "-DKNN_STAMP" knn.hip):
   EFFOCR_HIP_LIB=$PWD/tools/ab/lib_kstamp.so python tools/knn_timeline.py [D] [B] [N]
Prints, per workgroup (mean / p10 / p90 over the launch's workgroups, s_memtime ticks = 100 MHz reference clock units on gfx950's
s_memtime... the tool prints raw ticks and their share), the segments: query image, index stream (wave 0), wait for the other waves,
lists to LDS, list merge; plus the spread of workgroup start / end times over the launch."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effocr_amd.knn import IndexFlatIP

D = int(sys.argv[1]) if len(sys.argv) > 1 else 384
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
xb = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
ix = IndexFlatIP(D, device=dev, screen=False)
ix.add(xb)
q = torch.nn.functional.normalize(xb[:B] + 0.1 * torch.randn(B, D, generator=g, device=dev), dim=1)
for _ in range(5):
    ix.search_device(q, 10)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
    ix.search_device(q, 10)
ev[1].record()
torch.cuda.synchronize()
print(f"D={D} B={B} N={N}: {ev[0].elapsed_time(ev[1]) / 20 * 1e3:.1f} us per search (events around 20 calls)")
lib = ctypes.CDLL(os.environ["EFFOCR_HIP_LIB"])
NW, NS = 256, 8
buf = (ctypes.c_ulonglong * (NW * NS))()
assert lib.effocr_debug_knn_stamps(buf, NW * NS) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(NW, NS).astype(np.int64)
t = t[t[:, 5] > 0]
names = ["query image -> LDS", "index stream (wave 0)", "wait for the other waves", "lists -> LDS", "list merge (wave 0's queries)"]
tot = t[:, 5] - t[:, 0]
print(f"{len(t)} workgroups; ticks per workgroup mean {tot.mean():.0f} (min {tot.min()}, max {tot.max()})")
for i, nme in enumerate(names):
    seg = t[:, i + 1] - t[:, i]
    print(f"  {nme:32s} {seg.mean():9.0f}  {100 * seg.mean() / tot.mean():5.1f} %   (p10 {np.percentile(seg, 10):.0f}, p90 {np.percentile(seg, 90):.0f})")
s0, e0 = t[:, 0].min(), t[:, 5].max()
print(f"launch span (first start -> last end) {e0 - s0} ticks; starts spread {t[:, 0].max() - s0}; ends spread {e0 - t[:, 5].min()}; "
      f"ticks -> us needs the s_memtime rate: span / kernel time")
