#!/usr/bin/env python
"""EffRecognizer.run(numpy) — the call the reference's ONNX driver makes (infer_effocr_onnx_multi.py:161-163) — with 1 / 4 caller
threads over the staging knobs (lanes, copier threads, slices): crops/s per setting.  python tools/host_path_sweep.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from effocr_amd.recognizer_engine import EffRecognizer
from effocr_amd.weights import init_state_dict
arch = "vit_small_patch16_224"
sd = init_state_dict(arch, seed=0, img_size=224)
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
batches = [rng.standard_normal((64, 3, 224, 224), dtype=np.float32) for _ in range(8)]     # distinct arrays, like create_batches' list
print("cpu count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
# raw rates on this host
t0 = time.perf_counter(); dst = np.empty_like(batches[0])
for _ in range(10): np.copyto(dst, batches[0])
print(f"one-thread numpy copy: {10 * batches[0].nbytes / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
pin = torch.empty(batches[0].size, dtype=torch.float32).pin_memory(); d = torch.empty(batches[0].size, dtype=torch.float32, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): d.copy_(pin, non_blocking=True)
torch.cuda.synchronize(); print(f"pinned H2D: {10 * batches[0].nbytes / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
for lanes, copiers, slices, staging in ((2, 4, 8, "pinned"), (2, 4, 8, "direct"), (3, 4, 8, "direct"), (4, 4, 8, "direct"), (2, 4, 8, "pinned"), (2, 4, 8, "direct"), (4, 4, 8, "direct")):
    eng = EffRecognizer(sd, arch=arch, precision="bf16", device=dev, lanes=lanes, copiers=copiers, slices=slices, staging=staging)
    for b in batches[:2]: eng.run(b)
    res = []
    for nthr in (1, 4, 8):
        n_calls = 192
        def worker(i, n):
            for j in range(n): eng.run(batches[(i + j) % len(batches)])
        ths = [threading.Thread(target=worker, args=(i, n_calls // nthr)) for i in range(nthr)]
        t0 = time.perf_counter()
        for th in ths: th.start()
        for th in ths: th.join()
        res.append(64 * n_calls / (time.perf_counter() - t0))
    print(f"{staging:7s} lanes {lanes} copiers {copiers:2d} slices {slices:2d}: 1 thread {res[0]:8.0f}  4 threads {res[1]:8.0f}  8 threads {res[2]:8.0f} crops/s", flush=True)
    del eng
