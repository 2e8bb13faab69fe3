"""One library build (EFFOCR_HIP_LIB or the in-tree one): embedding error of the 16-bit modes against the library's exact-fp32 mode on 256 crops
(ViT-S; 64 for ViT-B) and the 1024-crop encoder time per mode.  Used by the round-4 f16 GELU-degree A/B (gpurun_out/f16ab.txt)."""
import sys, time, os
import torch
sys.path.insert(0, ".")
from effocr_amd.encoders import HipEncoder
from effocr_amd.weights import init_state_dict
dev = torch.device("cuda:0")
tag = os.environ.get("EFFOCR_HIP_LIB", "in-tree")
for arch, n_err in (("vit_small_patch16_224", 256), ("vit_base_patch16_224", 64)):
    sd = init_state_dict(arch, seed=0, img_size=224)
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(1024, 3, 224, 224, generator=g, device=dev)
    ref = HipEncoder(arch, sd, img_size=224, precision="fp32", device=dev).forward(x[:n_err], normalize=True)
    for prec in ("bf16", "fp16"):
        enc = HipEncoder(arch, sd, img_size=224, precision=prec, device=dev)
        e = enc.forward(x[:n_err], normalize=True)
        rel = ((e - ref).abs().max() / ref.abs().max()).item()
        rows = ((e - ref).abs().amax(1) / ref.abs().max()).sort().values
        for _ in range(3):
            enc.forward(x, normalize=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                enc.forward(x, normalize=True)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 10)
        print(f"{tag:40s} {arch:22s} {prec}: rel_err max {rel:.3e} median-row {rows[len(rows)//2].item():.3e}  {1e3*min(ts):.3f} ms per 1024 crops", flush=True)
        del enc
    torch.cuda.empty_cache()
