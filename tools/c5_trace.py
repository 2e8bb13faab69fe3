#!/usr/bin/env python
"""Kernel timeline of one run_effocr call (BASELINE configs[4] shape): run under rocprofv3 --kernel-trace, then summarise with --report.
   cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/c5trace -o t -- python $GRAFT_REPO_ROOT/tools/c5_trace.py [--bf16] [--host]
   python tools/c5_trace.py --report gpurun_out/c5trace/t_results.db"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def report(db):
    import sqlite3
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    # calls are separated by idle gaps > 1.5 ms (host-side sleep between calls)
    calls, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[1] - cur[-1][2] > 30e6:
            calls.append(cur); cur = []
        cur.append(r)
    calls.append(cur)
    last = calls[-1]
    t0 = last[0][1]
    print(f"{len(calls)} groups; last call: {len(last)} kernels, span {(last[-1][2] - t0) / 1e6:.3f} ms, busy {sum(r[2] - r[1] for r in last) / 1e6:.3f} ms")
    agg = {}
    prev_end = t0
    gaps = []
    for n, s, e in last:
        import re
        m = re.search(r"(\w+_kernel\w*|\w+kernel)\s*<([^>]{0,40})", n) or re.search(r"\d+(\w+?_kernel)I(\w{0,40})", n)
        k = (m.group(1) + "<" + m.group(2) + ">") if m else re.sub(r"\(anonymous namespace\)::", "", n.split("(")[0])[-70:]
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e6
        if s - prev_end > 20e3: gaps.append(((prev_end - t0) / 1e6, (s - prev_end) / 1e6, k))
        prev_end = max(prev_end, e)
    for k, (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"  {ms:8.3f} ms  x{cnt:4d}  {k}")
    print("idle gaps > 20 us (at ms, length ms, next kernel):")
    for g in gaps: print(f"  @{g[0]:8.3f}  {g[1]:7.3f}  {g[2]}")

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    report(sys.argv[2]); sys.exit(0)

import numpy as np, torch
from effocr_amd.knn import FaissKNN, IndexFlatIP
from effocr_amd.localizer_engine import EffLocalizer, init_yolov5s_state_dict
from effocr_amd.pipeline import run_effocr
from effocr_amd.recognizer_engine import EffRecognizer
from effocr_amd.transforms import PairedTransform
from effocr_amd.weights import init_state_dict
dev = torch.device("cuda:0")
nc = 2
sd = init_yolov5s_state_dict(nc, seed=0)
for l in range(3):
    b = sd[f"model.24.m.{l}.bias"].view(3, nc + 5); b[:, 4] += 5.5; b[:, 5] += 2.5
loc = EffLocalizer(sd, iou_thresh=0.05, conf_thresh=0.5, device=dev)
if "--bf16" in sys.argv: loc._eng_net.set_option("bf16_operands", 1)
arch = "vit_small_patch16_224"
rec = EffRecognizer(init_state_dict(arch, seed=0, img_size=224), arch=arch, precision="bf16", device=dev)
knn = FaissKNN(index_init_fn=IndexFlatIP, reset_before=False, reset_after=False, device=dev)
knn.train(torch.nn.functional.normalize(torch.randn(10000, 384, generator=torch.Generator().manual_seed(0)), dim=1))
chars = [chr(0x4E00 + i) for i in range(10000)]
tf = PairedTransform(size=224, device=dev)
rng = np.random.default_rng(0)
NL = int(sys.argv[sys.argv.index('--lines') + 1]) if '--lines' in sys.argv else 16
lines = [(rng.integers(0, 256, (256, 4096, 3)) // 32 * 32).astype(np.uint8) for _ in range(NL)]
inp = lines if "--host" in sys.argv else [torch.from_numpy(im).to(dev) for im in lines]
for i in range(5):
    torch.cuda.synchronize(); time.sleep(0.05)
    t0 = time.perf_counter()
    res, _ = run_effocr(inp, loc, rec, tf, "jp", knn_func=knn, candidate_chars=chars, max_det=1000 if '--lines' in sys.argv else 64, overlap_localizer='--serial' not in sys.argv)
    print(f"call {i}: {1e3 * (time.perf_counter() - t0):.2f} ms, {sum(len(v) for v in res.values())} chars", flush=True)
