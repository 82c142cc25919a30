"""Developer micro-benchmarks of individual C-ABI kernels at stage-0 shapes."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd import _lib
lib = _lib.load()
dev = 'cuda'
M, Cc = 311296, 40
H = 4 * Cc
bf = torch.bfloat16
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
h = torch.randn(M, H, device=dev, dtype=bf); z = torch.empty_like(h); dz = torch.randn_like(h)
scale = torch.rand(H, device=dev) + 0.5; beta = torch.randn(H, device=dev); coef = torch.randn(H, device=dev)
act = torch.ones(M, dtype=torch.uint8, device=dev)
for name, fn in [
    ("grn_apply act, out-of-place", lambda: lib.mpmae_grn_apply(1, P(h), P(z), P(scale), P(beta), M, H, M, P(act), st)),
    ("grn_apply no act", lambda: lib.mpmae_grn_apply(1, P(h), P(z), P(scale), P(beta), M, H, M, None, st)),
    ("grn_apply in place", lambda: lib.mpmae_grn_apply(1, P(h), P(h), P(scale), P(beta), M, H, M, None, st)),
    ("grn_bwd_apply", lambda: lib.mpmae_grn_bwd_apply(1, P(dz), P(h), P(scale), P(coef), M, H, M, st)),
    ("torch copy", lambda: z.copy_(h)),
]:
    us = t(fn)
    print(f"{name:32s} {us:8.1f} us")
