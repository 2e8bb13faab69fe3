"""BASELINE configs[1]'s own k-NN call (1024 queries x 10 000 rows x 384, k = 10): time of the screened and the exact search as a function of the
tile kernel's workgroup target (effocr_knn_set_option "wg_target": how finely the index is cut into chunks)."""
import sys, time, torch
sys.path.insert(0, ".")
from effocr_amd import _lib
from effocr_amd.knn import IndexFlatIP
dev = torch.device("cuda:0")
L = _lib.lib()
g = torch.Generator(device=dev).manual_seed(0)
for N, D, B, k in ((10000, 384, 1024, 10), (10000, 384, 128, 10), (10000, 384, 1024, 1), (96, 512, 64, 10)):
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
    Q = torch.nn.functional.normalize(torch.randn(B, D, generator=g, device=dev), dim=1)
    for target in (64, 128, 256, 512, 1024, 2048):
        _lib.check(L.effocr_knn_set_option(b"wg_target", target), "set")
        row = [f"N={N} D={D} B={B} k={k} wg_target={target:5d}"]
        for screen in (True, False):
            idx = IndexFlatIP(D, device=dev, screen=screen)
            idx.add(X)
            for _ in range(5):
                idx.search_device(Q, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                idx.search_device(Q, k)
            torch.cuda.synchronize()
            row.append(f"{'screened' if screen else 'exact'} {(time.perf_counter() - t0) / 50 * 1e6:7.1f} us")
        print("  ".join(row), flush=True)
_lib.check(L.effocr_knn_set_option(b"wg_target", 1024), "set")
