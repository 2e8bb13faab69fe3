#!/usr/bin/env python
"""Timeline of the SPLIT PARTS of the fused proj+MLP kernel in a small call (tools/ab_build.sh stamp_part "-DMLP_STAMP -DMLP_STAMP_PART=true" mlp_bf16p.hip):
   EFFOCR_HIP_LIB=$PWD/tools/ab/lib_stamp_part.so python tools/mlp_part_timeline.py [crops]
Every split part (wave 0) stamps s_memtime at its milestones; prints the mean ticks per segment, the span of the launch inside an
XCD (first entry -> last exit) and the dispatch ramp (entry of the k-th workgroup of an XCD after the first)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effocr_amd.encoders import HipEncoder
from effocr_amd.weights import init_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
enc = HipEncoder("vit_small_patch16_224", init_state_dict("vit_small_patch16_224", seed=0, img_size=224), img_size=224, precision="bf16", device=dev)
enc.set_option("cls_only_last", 0); enc.set_option("pair_parts", 0); enc.set_option("mlp_pair", -1)   # the 128-token split parts             # every block's launch has the same shape: the last one's stamps are those of a middle block
x = torch.randn(B, 3, 224, 224, device=dev)
for _ in range(5):
    enc.forward(x, normalize=True)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["EFFOCR_HIP_LIB"])
NW, NS = 2048, 20
buf = (ctypes.c_ulonglong * (NW * NS))()
# (calls of <= 36 crops run "pair parts" — kernels of mlp_bf16pair.hip, whose stamp table is read by effocr_debug_mlp_pair_stamps; this tool measures the
#  128-token parts: pair_parts = 0, mlp_pair = -1 above)
rc = lib.effocr_debug_mlp_stamps(buf, NW * NS)
assert rc == 0, rc
t = np.frombuffer(buf, dtype=np.uint64).reshape(NW, NS).astype(np.int64)
npan = (B * 197 + 127) // 128
split = 6 if npan * 6 <= 256 else 4 if npan * 4 <= 256 else 2
nw = npan * split
t = t[:nw]
names = {1: "entry", 2: "requests issued + params in regs", 3: "params -> LDS + barrier", 4: "rows / attn / stage 0 landed", 5: "projection MFMAs", 6: "bias + LayerNorm (+ zero)",
         9: "hidden chunks (A / hand-over / B)", 11: "partial stores issued", 12: "stores acked"}
# stamp indices: 0 entry, 1 (after stagger: none here), 2 params to LDS, 3 stage 0 landed, 4 projection done, 5 LayerNorm done, 8 chunks done, 10 stores issued, 11 stores acked
order = [0, 2, 3, 4, 5, 8, 10, 11]
label = ["requests issued, params -> LDS, barrier", "rows / attention fragments / ring stage 0 landed", "projection (18 stages)", "bias + LayerNorm",
         f"{12 // split} hidden chunks", "partial stores issued", "stores acked"]
tot = t[:, 11] - t[:, 0]
print(f"{B} crops: {npan} panels x {split} parts = {nw} workgroups; ticks per part mean {tot.mean():.0f} (min {tot.min()}, max {tot.max()})  [100 MHz ticks x 10 ns if < 10 k, else shader clocks]")
for i in range(1, len(order)):
    seg = t[:, order[i]] - t[:, order[i - 1]]
    print(f"  {label[i - 1]:52s} {seg.mean():9.0f}  {100 * seg.mean() / tot.mean():5.1f} %   (p10 {np.percentile(seg, 10):.0f}, p90 {np.percentile(seg, 90):.0f})")
print(f"  mid-stage waits of wave 0, sum over the part's stages: vmcnt {t[:, 12].mean():.0f}, barrier {t[:, 13].mean():.0f}")
rt0, rt1 = t[:, 14], t[:, 16]                   # 100 MHz real-time counter (one domain for the whole chip) at entry / exit
e = np.sort(rt0) - rt0.min()
print(f"  launch, real-time counter (10 ns ticks): first entry -> last exit {(rt1.max() - rt0.min()) / 100:.2f} us; entries after the first: median {np.median(e) / 100:.2f} us, "
      f"p90 {np.percentile(e, 90) / 100:.2f}, last {e[-1] / 100:.2f} us; a part entry -> exit mean {(rt1 - rt0).mean() / 100:.2f} us (max {(rt1 - rt0).max() / 100:.2f})")
print(f"  => shader clock over a part {tot.mean() / ((rt1 - rt0).mean() * 10):.2f} GHz" if (rt1 - rt0).mean() > 0 else "")
