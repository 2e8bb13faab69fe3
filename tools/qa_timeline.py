"""Per-phase timeline of the fused qkv+attention kernel (build with EXP=3064): python tools/qa_timeline.py"""
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
buf = np.zeros(1024 * 32, dtype=np.uint64)
L.effocr_exp_qa_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.effocr_exp_qa_timeline(buf.ctypes.data, buf.size) == 0
t = buf.reshape(1024, 32).astype(np.int64)[:256]
d = np.diff(t[:, :19], axis=1)
print('last image of each workgroup; total ticks per image:', np.median(t[:, 18] - t[:, 0]))
lab = sum([[f'head slot {h} proj', f'slot {h} kv barrier', f'slot {h} attention'] for h in range(6)], [])
for i, n in enumerate(lab):
    print(f'{n:30s} {np.median(d[:, i]):10.0f}  (p10 {np.percentile(d[:, i], 10):8.0f}, p90 {np.percentile(d[:, i], 90):8.0f})')
