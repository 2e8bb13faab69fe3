# needs an experiment build of the library (make -C effocr_amd/csrc EXP=1): effocr_dbg_ln_linear is compiled out of the shipped .so
"""experiment: timeline of two workgroups of the panel kernel (s_memtime stamps; build with EXP=9)."""
import ctypes, sys, math, torch, statistics
from effocr_amd import _lib
L = _lib.lib()
L.effocr_dbg_ln_linear.restype = ctypes.c_int
L.effocr_dbg_ln_linear.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
M, N, K = 1024 * 197, int(sys.argv[1]) if len(sys.argv) > 1 else 1536, 384
epi = int(sys.argv[2]) if len(sys.argv) > 2 else 1
x = torch.randn(M, K, device=dev); g = torch.ones(K, device=dev); b = torch.zeros(K, device=dev)
w = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16(); bias = torch.zeros(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
dbg = torch.zeros(4 * 2048, dtype=torch.int64, device=dev)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    dbg.zero_()
    ev0.record()
    rc = L.effocr_dbg_ln_linear(0, epi, _lib.ptr(x), _lib.ptr(g), _lib.ptr(b), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(out), M, N, K, 0, _lib.ptr(dbg), None)
    ev1.record()
    assert rc == 0, L.effocr_last_error()
    torch.cuda.synchronize()
print("kernel ms", ev0.elapsed_time(ev1))
d = dbg.cpu().view(4, 2048)
S = (N // 128) * 6
for wsel, name in enumerate(["wg300/w0", "wg300/w5", "wg1300/w0", "wg1300/w5"]):
    t = d[wsel]
    st = t[:S * 4].view(S, 4)
    p0, p1, p2, pend = [t[2040 + i].item() for i in range(4)]
    if p0 == 0: print(name, "no data"); continue
    print(f"{name}: prologue(load+LN) {p1-p0}  sync {p2-p1}  loop {st[S-1,3].item()-st[0,0].item()} ({S} stages)  tail {pend-st[S-1,3].item()}  total {pend-p0}")
    rows = [(s, (st[s,1]-st[s,0]).item(), (st[s,2]-st[s,1]).item(), (st[s,3]-st[s,2]).item(), (st[s+1,0]-st[s,3]).item() if s+1<S else 0) for s in range(S)]
    print("  stage vmcnt barrier body gap :", " | ".join("%d:%d/%d/%d/%d" % r for r in rows[:14]))
    print("  mean vmcnt %.0f barrier %.0f body %.0f gap %.0f" % tuple(statistics.mean(r[i] for r in rows) for i in (1,2,3,4)))
