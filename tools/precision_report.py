import sys, numpy as np, torch
sys.path.insert(0, '.')
from effocr_amd.encoders import HipEncoder
from effocr_amd.weights import init_state_dict
dev = torch.device("cuda:0")
for arch in ["vit_small_patch16_224", "vit_base_patch16_224"]:
    g = np.load(f"tests/golden/enc_{arch}.npz")
    sd = init_state_dict(arch, seed=int(g["seed"]), img_size=int(g["img"]))
    x = torch.from_numpy(g["x"].astype(np.float32)).to(dev)
    for prec in ["fp32", "fp16", "bf16"]:
        emb = HipEncoder(arch, sd, img_size=int(g["img"]), precision=prec, device=dev).forward(x).cpu().numpy()
        ref = g["emb"]
        cos = (emb * ref).sum(1) / np.linalg.norm(emb, axis=1) / np.linalg.norm(ref, axis=1)
        print(f"{arch:24s} {prec}: max-abs/max-abs {np.abs(emb - ref).max() / np.abs(ref).max():.2e}  min cosine {cos.min():.6f}")
