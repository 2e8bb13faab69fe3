"""Per-phase timeline of the fused proj+MLP kernel (build with EXP=5072: MLX bits 1024 (+ the unused 2048)): python tools/mlp_timeline.py"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, '.')
from effocr_amd import _lib
from effocr_amd.encoders import HipEncoder
from effocr_amd.weights import init_state_dict
dev = torch.device('cuda:0')
sd = init_state_dict('vit_small_patch16_224', seed=0, img_size=224)
enc = HipEncoder('vit_small_patch16_224', sd, img_size=224, precision='bf16', device=dev)
x = torch.randn(1024, 3, 224, 224, device=dev)
for _ in range(2):
    enc.forward(x, normalize=True)
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(2048 * 16, dtype=np.uint64)
L.effocr_exp_mlp_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.effocr_exp_mlp_timeline(buf.ctypes.data, buf.size) == 0
t = buf.reshape(2048, 16).astype(np.int64)[256:1536]
d = np.diff(t[:, :9], axis=1)
tot = t[:, 8] - t[:, 0]
print('ticks per panel (median):', np.median(tot), ' first-to-last start spread:', t[:, 0].max() - t[:, 0].min())
lab = ['params + fragment loads', 'projection (288 MFMAs)', 'x loads + in-place add', 'LayerNorm', 'A(0) + park', 'rolled loop (10 chunks)',
       'last A/B phases (2 chunks of B, 1 of A)', 'epilogue']
for i, n in enumerate(lab):
    print(f'{n:42s} {np.median(d[:, i]):9.0f} {100 * np.median(d[:, i]) / np.median(tot):5.1f} %   (p10 {np.percentile(d[:, i], 10):8.0f}, p90 {np.percentile(d[:, i], 90):8.0f})')
