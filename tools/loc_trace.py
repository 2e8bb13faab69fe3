#!/usr/bin/env python
"""Per-launch timeline of ONE YOLOv5s localizer forward (16 images of 640 x 640): run under rocprofv3 --kernel-trace, then --report.
   cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/loctrace -o t -- python $GRAFT_REPO_ROOT/tools/loc_trace.py
   python tools/loc_trace.py --report gpurun_out/loctrace/<host>/<pid>_results.db"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    import sqlite3, re
    c = sqlite3.connect(sys.argv[2])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    rows = c.execute(f"select name, start, end{', ' + gx if gx else ''} from kernels order by start").fetchall()
    # the last forward = the last run of kernels after the final long idle gap
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][1] - rows[i - 1][2] > 20e6: cut = i
    last = rows[cut:]
    tot = sum(r[2] - r[1] for r in last) / 1e3
    print(f"columns {cols}")
    print(f"last forward: {len(last)} launches, busy {tot:.1f} us, span {(last[-1][2] - last[0][1]) / 1e3:.1f} us")
    for r in last:
        n = re.sub(r"\(anonymous namespace\)::|effocr::", "", r[0])[:60]
        print(f"  {(r[2] - r[1]) / 1e3:8.1f} us  grid {r[3] if gx else '?':>8}  {n}")
    sys.exit(0)
import time, torch
from effocr_amd.localizer_engine import HipLocalizer, init_yolov5s_state_dict
dev = torch.device("cuda:0")
loc = HipLocalizer(init_yolov5s_state_dict(2, seed=0), input_shape=(640, 640), device=dev)
im = torch.rand(16, 3, 640, 640, device=dev)
for _ in range(3):
    loc.forward(im); torch.cuda.synchronize(); time.sleep(0.05)
