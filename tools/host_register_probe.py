#!/usr/bin/env python
"""hipHostRegister of a caller-owned 64-crop batch (38.5 MB) vs the staging copy: register + H2D + unregister timings."""
import time, numpy as np, torch
dev = torch.device("cuda:0")
rt = torch.cuda.cudart()
rng = np.random.default_rng(0)
arrs = [rng.standard_normal((64, 3, 224, 224), dtype=np.float32) for _ in range(6)]
d = torch.empty(arrs[0].shape, dtype=torch.float32, device=dev)
pin = torch.empty(arrs[0].shape, dtype=torch.float32).pin_memory()
torch.cuda.synchronize()
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"pinned -> device (38.5 MB): {t(lambda: d.copy_(pin, non_blocking=True)):.3f} ms")
i = [0]
def staged():
    a = arrs[i[0] % 6]; i[0] += 1
    np.copyto(pin.numpy(), a); d.copy_(pin, non_blocking=True)
print(f"copyto pinned + H2D (one thread): {t(staged):.3f} ms")
def pageable():
    a = arrs[i[0] % 6]; i[0] += 1
    d.copy_(torch.from_numpy(a))
print(f"pageable -> device (torch copy_): {t(pageable):.3f} ms")
def registered():
    a = arrs[i[0] % 6]; i[0] += 1
    r = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)
    assert int(r) == 0, r
    d.copy_(torch.from_numpy(a), non_blocking=True)
    torch.cuda.synchronize()
    rt.cudaHostUnregister(a.ctypes.data)
print(f"register + H2D + sync + unregister: {t(registered):.3f} ms")
a = arrs[0]
t0 = time.perf_counter(); rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0); t1 = time.perf_counter(); rt.cudaHostUnregister(a.ctypes.data); t2 = time.perf_counter()
print(f"register {1e3 * (t1 - t0):.3f} ms, unregister {1e3 * (t2 - t1):.3f} ms")
