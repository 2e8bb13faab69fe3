#!/usr/bin/env python
"""Per-panel timeline of the fused proj+MLP kernel from a -DMLP_STAMP build (tools/ab_build.sh stamp "-DMLP_STAMP" mlp_bf16p.hip mlp_bf16pair.hip):
   EFFOCR_HIP_LIB=$PWD/tools/ab/lib_stamp.so python tools/mlp_timeline.py [batch]
Prints the mean share of each segment of a whole panel (s_memtime ticks, wave 0) and the per-CU turn-around between workgroups."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effocr_amd import _lib
from effocr_amd.encoders import HipEncoder
from effocr_amd.weights import init_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
PAIR = "--pair" in sys.argv                    # 64-token wave-pair panels (calls of <= 83 crops): every workgroup of the launch stamps
dev = torch.device("cuda:0")
enc = HipEncoder("vit_small_patch16_224", init_state_dict("vit_small_patch16_224", seed=0, img_size=224), img_size=224, precision="bf16", device=dev)
if PAIR:
    enc.set_option("mlp_pair", 1); enc.set_option("cls_only_last", 0)
x = torch.randn(B, 3, 224, 224, device=dev)
for _ in range(3):
    enc.forward(x, normalize=True)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["EFFOCR_HIP_LIB"])
NW, NS = 2048, 20
buf = (ctypes.c_ulonglong * (NW * NS))()
rc = (lib.effocr_debug_mlp_pair_stamps if PAIR else lib.effocr_debug_mlp_stamps)(buf, NW * NS)   # (the pair kernels live in a translation unit of their own: its table)
assert rc == 0, rc
t = np.frombuffer(buf, dtype=np.uint64).reshape(NW, NS).astype(np.int64)
npan = (B * 197 + 127) // 128
main = npan - npan % 256 if (npan % 256) * 2 <= 256 else npan
t = t[:(B * 197 + 63) // 64] if PAIR else t[8:main]   # (the CLS-only launch of the last block rewrote the first 8 rows)
names = ["stagger wait", "requests issued + params to regs", "params -> LDS + barrier", "rows/attn/stage 0 landed", "projection MFMAs", "bias + LayerNorm",
         "A(0) + park", "rolled loop", "last A/B + gelu + B", "x stores issued", "second output", "stores acked"]
tot = t[:, 11] - t[:, 1]
print(f"{len(t)} whole panels; ticks per panel (after the start spread) mean {tot.mean():.0f} (min {tot.min()}, max {tot.max()}); start spread mean {(t[:, 1] - t[:, 0]).mean():.0f}")
for i in range(2, 12):
    seg = t[:, i] - t[:, i - 1]
    print(f"  {names[i]:42s} {seg.mean():9.0f}  {100 * seg.mean() / tot.mean():5.1f} %   (p10 {np.percentile(seg, 10):.0f}, p90 {np.percentile(seg, 90):.0f})")
print(f"  projection by output group (stamps 17, 18): group 0 {(t[:, 17] - t[:, 3]).mean():.0f}, group 1 {(t[:, 18] - t[:, 17]).mean():.0f}, group 2 {(t[:, 4] - t[:, 18]).mean():.0f} ticks (96 MFMAs = 3 072 each)")
print(f"  mid-stage waits of one wave, sum over the panel's {12 * 12 + 18} stages: vmcnt {t[:, 12].mean():.0f} ticks, barrier {t[:, 13].mean():.0f} ticks")
xcc = (t[:, 15] >> 32) & 0xf                     # s_memtime is per XCD: concurrency inside each XCD (32 CUs), then averaged
cm, mx, fr = [], [], []
for xc in np.unique(xcc):
    tt = t[xcc == xc]
    ev = np.concatenate([np.stack([tt[:, 1], np.ones(len(tt))], 1), np.stack([tt[:, 2], -np.ones(len(tt))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    conc = np.cumsum(ev[:, 1]); dt = np.diff(ev[:, 0])
    cm.append((conc[:-1] * dt).sum() / dt.sum()); mx.append(conc.max()); fr.append(len(tt))
print(f"  workgroups in the prologue wait at once, per XCD of 32 CUs (time-weighted over the launch): mean {np.mean(cm):.1f}, max {np.max(mx):.0f}  (XCDs seen {len(cm)}, panels per XCD {np.mean(fr):.0f})")
# per-CU turn-around: group by (xcc, se, cu) and sort by start
hw = t[:, 15]
cu = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 0x7) * 256 + ((hw >> 8) & 0xf) * 16 + ((hw >> 12) & 1)
gaps = []
for c in np.unique(cu):
    rows = t[cu == c]
    rows = rows[np.argsort(rows[:, 0])]
    gaps += list(rows[1:, 0] - rows[:-1, 11])
gaps = np.array(gaps)
print(f"CUs seen {len(np.unique(cu))}; turn-around end(stores acked) -> next workgroup's entry on the same CU: mean {gaps.mean():.0f}, p10 {np.percentile(gaps, 10):.0f}, p90 {np.percentile(gaps, 90):.0f} ticks")
