import sys, torch
sys.path.insert(0, '.')
from effocr_amd.knn import IndexFlatIP
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)
for N, D in [(1_000_000, 768), (100_000, 384), (10_000, 384)]:
    X = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
    Q = torch.nn.functional.normalize(X[torch.randint(0, N, (1024,), generator=g, device=dev)] + 0.1 * torch.randn(1024, D, generator=g, device=dev), dim=1)
    res = {}
    for name, scr in [("exact", False), ("screened", True)]:
        idx = IndexFlatIP(D, device=dev, screen=scr); idx.add(X)
        for _ in range(2): out = idx.search_device(Q, 10)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): out = idx.search_device(Q, 10)
        e1.record(); torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 5, out)
    same = torch.equal(res["exact"][1][1], res["screened"][1][1]) and torch.equal(res["exact"][1][0].view(torch.int32), res["screened"][1][0].view(torch.int32))
    print(f"N={N:8d} D={D}: exact {res['exact'][0]:7.3f} ms  screened {res['screened'][0]:7.3f} ms  identical={same}")
