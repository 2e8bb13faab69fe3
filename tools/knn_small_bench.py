"""k-NN at small batch sizes against a 1M-row index (HBM-bound regime): python tools/knn_small_bench.py [D ...]"""
import sys, time
import torch
sys.path.insert(0, ".")
from effocr_amd.knn import IndexFlatIP
dev = torch.device("cuda:0")
for D in [int(v) for v in sys.argv[1:]] or [384, 768]:
    N = 1000000
    idx = IndexFlatIP(D, device=dev, screen=False)
    g = torch.Generator(device=dev).manual_seed(1)
    idx.add(torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1))
    for B in (1, 16, 32):
        q = idx._xb[:B].clone()
        for _ in range(3):
            idx.search_device(q, 10)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            idx.search_device(q, 10)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
        print(f"D={D} B={B}: {t*1e3:.3f} ms  {N*D*4/t/1e12:.2f} TB/s  {N*D*4/t/8e12:.3f} of 8 TB/s")
    del idx
