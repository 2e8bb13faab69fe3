"""Yardstick only (not product): vendor-library bf16 GEMM rate on the encoder's four linear shapes."""
import torch
dev = torch.device("cuda:0")
M = 1024 * 197
for name, N, K in [("qkv", 1152, 384), ("proj", 384, 384), ("fc1", 1536, 384), ("fc2", 384, 1536),
                   ("vitb_qkv", 2304, 768), ("vitb_fc1", 3072, 768), ("vitb_fc2", 768, 3072)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    for label, fn in [("mm", lambda: torch.mm(x, w.t())), ("addmm(bias)", lambda: torch.addmm(b, x, w.t()))]:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"{name:9s} {label:12s} M={M} N={N:5d} K={K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
