"""f16 / bf16 embedding error of ViT-S against the library's fp32 mode (and fp32 mode against oracle A) as the synthetic weights are pushed to
larger activation magnitudes (tests/test_gpu_encoder.py::_trained_magnitudes): how much of the error is range, how much is conditioning."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from effocr_amd.encoders import HipEncoder
from test_gpu_encoder import _trained_magnitudes, rel_err
from oracle.encoders_ref import encoder_forward, l2_normalize
dev = torch.device("cuda:0")
arch, img = "vit_small_patch16_224", 224
x = torch.randn(24, 3, img, img, generator=torch.Generator().manual_seed(6))
for resid, qg, fg in [(1, 1, 1), (64, 1, 1), (1, 6, 1), (1, 1, 400), (1, 1, 100), (64, 3, 100), (64, 6, 100), (64, 3, 200), (64, 6, 400), (64, 2, 50)]:
    sd = _trained_magnitudes(arch, img, seed=5, resid=float(resid), q_gain=float(qg), fc1_gain=float(fg))
    ref = HipEncoder(arch, sd, img_size=img, precision="fp32", device=dev).forward(x.to(dev), normalize=True).cpu()
    ora = l2_normalize(encoder_forward(arch, sd, x[:4]))
    row = [f"resid x{resid:<3} norm1 x{qg:<2} norm2 x{fg:<4} fp32-vs-oracle {rel_err(ref[:4], ora):.2e}"]
    for prec in ("fp16", "bf16"):
        e = HipEncoder(arch, sd, img_size=img, precision=prec, device=dev)
        got = e.forward(x.to(dev), normalize=True).cpu()
        row.append(f"{prec} {rel_err(got, ref):.2e} finite={bool(torch.isfinite(got).all())}")
    print("  ".join(row), flush=True)
