#!/usr/bin/env python
"""Launch table of ONE 64-crop and ONE 128-crop recognizer call (encoder + fused L2 normalise + k-NN, k = 10, 10 000-row index; the
reference's real call sizes: infer_effocr_onnx_multi.py:157 pads every batch to 64, BASELINE configs[2] shards 1024 crops into 128 per
rank).  Run under rocprofv3 --kernel-trace, then summarise with --report:
   cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/sc_trace -o t -- python $GRAFT_REPO_ROOT/tools/small_call_trace.py [--precision bf16]
   python tools/small_call_trace.py --report gpurun_out/sc_trace/<host>/t_results.db > profiles/rNN_small_call_kernels.txt
Calls are separated by host sleeps; the report takes the LAST call of each size (everything warm) and lists every launch with the idle
gap in front of it."""
import os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN6effocr12_GLOBAL__N_1\d+", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"effocr::", "", n)
    return n.split("(")[0][:64]


def report(db):
    import sqlite3
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    groups, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[1] - cur[-1][2] > 20e6:                       # > 20 ms idle: the host sleep between calls
            groups.append(cur); cur = []
        cur.append(r)
    groups.append(cur)
    # the script prints markers in order: ... 64-crop calls, then 128-crop calls; the last group of each half
    n64 = int(os.environ.get("SC_CALLS", "6"))
    sizes = [int(v) for v in os.environ.get("SC_SIZES", "64,128").split(",")]
    calls = groups[-len(sizes) * n64:]
    for label, g in [(f"{B} crops", calls[(j + 1) * n64 - 1]) for j, B in enumerate(sizes)]:
        t0 = g[0][1]
        span, busy = (g[-1][2] - t0) / 1e3, sum(e - s for _, s, e in g) / 1e3
        print(f"== one call of {label}: {len(g)} launches, span {span:.1f} us, kernels busy {busy:.1f} us, idle {span - busy:.1f} us")
        print(f"{'#':>3s} {'start_us':>9s} {'gap_us':>7s} {'dur_us':>8s}  kernel")
        prev = t0
        agg = {}
        for i, (n, s, e) in enumerate(g):
            k = short(n)
            print(f"{i:3d} {(s - t0) / 1e3:9.1f} {(s - prev) / 1e3:7.1f} {(e - s) / 1e3:8.1f}  {k}")
            a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3; a[2] += max(0.0, (s - prev) / 1e3)
            prev = max(prev, e)
        print("   by kernel (launches, busy us, idle us in front):")
        for k, (cnt, b, gp) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"     x{cnt:3d} {b:8.1f} {gp:7.1f}  {k}")
        print()


if len(sys.argv) > 2 and sys.argv[1] == "--report":
    report(sys.argv[2]); sys.exit(0)

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
n_calls = int(os.environ.get("SC_CALLS", "6"))
for B in [int(v) for v in os.environ.get("SC_SIZES", "64,128").split(",")]:
    x = torch.randn(B, 3, 224, 224, device=dev)
    for i in range(n_calls):
        torch.cuda.synchronize(); time.sleep(0.06)
        t0 = time.perf_counter()
        emb = enc.forward(x, normalize=True)
        d, ids = idx.search_device(emb, 10)
        torch.cuda.synchronize()
        print(f"{B} crops, call {i}: {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
