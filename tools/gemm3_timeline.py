#!/usr/bin/env python
"""Per-tile timeline of the persistent gemm3 kernel from a -DGEMM3_TRACE build (tools/ab_build.sh g3trace "-DGEMM3_TRACE" gemm3.hip):
   EFFOCR_HIP_LIB=$PWD/tools/ab/lib_g3trace.so python tools/gemm3_timeline.py
Runs the four ViT-B linears (201 728 tokens) through effocr_op_linear_blocked (random data in the blocked layout: timing only) and prints
the mean length of each segment of a tile (s_memrealtime, 100 MHz, wave 0 of every workgroup)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effocr_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
lib = ctypes.CDLL(os.environ["EFFOCR_HIP_LIB"])
M = 1024 * 197
ra = (M + 127) // 128 * 128
WGS, TILES, NS = 256, 48, 10
names = ["tile-top barrier + constants DMA issue", "stages 0-2 (no waits)", "stage 3 (first counted wait: store acks)", "stages 4 .. nst-1",
         "ring wait + first chunk (loads / GELU)", "rest of the epilogue (to the last store issued)"]
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for name, N, K, epi in [("qkv", 2304, 768, "bias"), ("fc1", 3072, 768, "bias_gelu"), ("fc2", 768, 3072, "bias_resid"), ("proj", 768, 768, "bias_resid")]:
    x = torch.randn(ra * K // 2, device=dev).to(torch.bfloat16).repeat(2) * 0.5
    w = torch.randn(N * K, device=dev).to(torch.bfloat16) * (K ** -0.5)
    b = torch.randn(N, device=dev)
    if epi == "bias_resid":
        out = torch.randn(ra * N, device=dev); rd = out
    else:
        out = torch.zeros(ra * N, dtype=torch.bfloat16, device=dev); rd = None
    for _ in range(3):
        _lib.check(L.effocr_op_linear_blocked(_lib.PREC["bf16"], _lib.EPI[epi], _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(rd), _lib.ptr(out),
                                              M, N, K, ra, stream), "op_linear_blocked")
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    _lib.check(L.effocr_op_linear_blocked(_lib.PREC["bf16"], _lib.EPI[epi], _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(rd), _lib.ptr(out),
                                          M, N, K, ra, stream), "op_linear_blocked")
    ev1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (WGS * TILES * NS))()
    assert lib.effocr_debug_gemm3_stamps(buf, WGS * TILES * NS) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(WGS, TILES, NS).astype(np.int64)
    ntile = (M // 256) * (N // 256)
    per = min(ntile // 256, TILES)                      # whole rounds recorded (the tail launch, if any, overwrote tile 0 of some rows)
    tt = t[:, 1:per, :]                                  # skip each workgroup's first tile (cold ring, possibly overwritten)
    seg = np.diff(tt[:, :, :7], axis=2) / 100.0          # us
    tot = (tt[:, :, 6] - tt[:, :, 0]) / 100.0
    gap = (tt[:, 1:, 0] - tt[:, :-1, 6]) / 100.0         # last store issued -> next tile top
    print(f"== {name}: N {N} K {K} {epi}: launch {ev0.elapsed_time(ev1) * 1e3:.0f} us, {ntile} tiles = {ntile / 256:.2f} per workgroup; tile {tot.mean():.2f} us (p10 {np.percentile(tot, 10):.2f}, p90 {np.percentile(tot, 90):.2f}) + {gap.mean():.2f} us to the next tile top")
    for i, n in enumerate(names):
        sgi = seg[:, :, i]
        print(f"   {n:52s} {sgi.mean():7.2f} us  {100 * sgi.mean() / (tot.mean() + gap.mean()):5.1f} %   (p10 {np.percentile(sgi, 10):.2f}, p90 {np.percentile(sgi, 90):.2f})")
    clk = np.diff(tt[:, :, 7], axis=1) / np.diff(tt[:, :, 0], axis=1) * 0.1   # shader clocks (s_memtime) per 100 MHz tick -> GHz
    print(f"   shader clock over a tile: mean {clk.mean():.3f} GHz (p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f})")
    cyc = np.diff(tt[:, :, 7], axis=1).mean()                                 # shader ticks per tile
    print(f"   of {cyc:.0f} shader ticks per tile (wave 0): {tt[:, :, 8].mean():.0f} in the stages' vmcnt / lgkmcnt waits, {tt[:, :, 9].mean():.0f} at their barriers; MFMA issue minimum {K // 32 * 32 * 32} ticks")
    # how far apart are the workgroups?  start of tile k across workgroups
    sp = tt[:, :, 0].std(axis=0) / 100.0
    print(f"   spread (std over workgroups) of the tile-top time: first recorded tile {sp[0]:.2f} us, last {sp[-1]:.2f} us")
