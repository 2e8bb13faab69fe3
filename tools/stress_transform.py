import sys, numpy as np, torch
sys.path.insert(0, '.')
from effocr_amd.transforms import PairedTransform, slice_boxes
from oracle import crop_transform_ref as R
dev = torch.device("cuda:0")
rng = np.random.default_rng(123)
worst = {True: 0.0, False: 0.0}
bad = 0
for trial in range(30):
    H, W = int(rng.integers(8, 700)), int(rng.integers(8, 900))
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    boxes = []
    for _ in range(40):
        x0 = int(rng.integers(-5, W)); y0 = int(rng.integers(-5, H))
        w = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 100, 223, 224, 225, 447, 448, 449, 600])); h = int(rng.choice([1, 2, 5, 16, 57, 224, 300, 512]))
        boxes.append((x0, y0, x0 + w, y0 + h))
    ok = []
    for b in boxes:
        try:
            slice_boxes([b], H, W); ok.append(b)
        except ValueError:
            pass
    if not ok: continue
    for aa in (True, False):
        for size in (224, 32):
            got = PairedTransform(size=size, antialias=aa, device=dev).boxes(img, ok, already_int=True).cpu().numpy()
            ref = R.transform_boxes(img, ok, size=size, antialias=aa)
            ib = slice_boxes(ok, H, W)
            for i in range(len(ok)):
                L = max(ib[i, 2] - ib[i, 0], ib[i, 3] - ib[i, 1])
                tol = 2e-5 if aa else max(2e-5, 2e-6 * L) * 4.5
                e = float(np.abs(got[i] - ref[i]).max())
                worst[aa] = max(worst[aa], e / tol)
                if e > tol:
                    bad += 1
                    print("MISMATCH", (H, W), ok[i], aa, size, e, tol)
print("worst error / tolerance:", worst, "mismatches:", bad)
