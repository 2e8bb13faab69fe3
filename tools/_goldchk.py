import sys, os, numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, root)
from effocr_amd.encoders import HipEncoder
from effocr_amd.weights import init_state_dict
dev = torch.device("cuda:0")
arch = "vit_small_patch16_224"
g = np.load(os.path.join(root, "tests/golden", f"enc_{arch}.npz"))
sd = init_state_dict(arch, seed=int(g["seed"]), img_size=int(g["img"]))
x = torch.from_numpy(g["x"].astype(np.float32)).to(dev)
for prec in ("bf16", "fp16"):
    for opts in ({}, {"tail_split": 0}, {"split6": 0}, {"mlp_pair": 1}):
        enc = HipEncoder(arch, sd, img_size=int(g["img"]), precision=prec, device=dev)
        for k, v in opts.items(): enc.set_option(k, v)
        emb = enc.forward(x).cpu().numpy()
        e = np.abs(emb - g["emb"]); r = g["emb"]
        print(prec, opts, f"max-norm {e.max() / np.abs(r).max():.3e}  per-row L2 {(np.linalg.norm(e, axis=1) / np.linalg.norm(r, axis=1)).max():.3e}", flush=True)
