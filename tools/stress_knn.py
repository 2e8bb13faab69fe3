"""Ad-hoc stress (not a test): screened k-NN vs the exact kernel must be bit-identical on adversarial score distributions."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from effocr_amd.knn import IndexFlatIP
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(99)
bad = 0
for trial in range(24):
    D = [64, 128, 384, 768][trial % 4]
    N = int(torch.randint(66_000, 260_000, (1,)).item())
    X = torch.randn(N, D, generator=g, device=dev)
    mode = trial % 6
    if mode in (0, 1):
        X = torch.nn.functional.normalize(X, dim=1)
    if mode in (2, 3):                                   # mixed norms over 3 orders of magnitude
        X = X * (10 ** (torch.rand(N, 1, generator=g, device=dev) * 3 - 2))
    if mode in (1, 3, 4, 5):                             # clusters of near-duplicates at several tightness scales
        for c in range(12):
            centre = X[int(torch.randint(0, N, (1,)).item())].clone()
            size = [5, 40, 300, 700, 1500][c % 5]
            lo = int(torch.randint(0, N - size, (1,)).item())
            X[lo:lo + size] = centre + (10.0 ** -(1 + c % 5)) * centre.norm() / D ** 0.5 * torch.randn(size, D, generator=g, device=dev)
    B = [1, 37, 256, 1024][trial % 4]
    pick = torch.randint(0, N, (B,), generator=g, device=dev)
    Q = X[pick] + 0.02 * X[pick].norm(dim=1, keepdim=True) / D ** 0.5 * torch.randn(B, D, generator=g, device=dev)
    k = [1, 10, 16, 32][trial % 4]
    ex = IndexFlatIP(D, device=dev, screen=False); ex.add(X)
    sc = IndexFlatIP(D, device=dev, screen=True); sc.add(X)
    De, Ie = ex.search_device(Q, k); Ds, Is = sc.search_device(Q, k)
    torch.cuda.synchronize()
    same = torch.equal(Ie, Is) and torch.equal(De.view(torch.int32), Ds.view(torch.int32))
    bad += (not same)
    print(f"trial {trial:2d} N={N:7d} D={D:3d} B={B:4d} k={k:2d} mode={mode} identical={same}")
print("mismatching trials:", bad)
