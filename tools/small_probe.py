"""Developer timing of the small row-wise kernels at the decoder shape (M = 12 544 rows, C = 512), back to back on one stream."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd import _lib
lib = _lib.load()
dev = "cuda"
for kv in os.environ.get("LIBOPTS", "").split(","):
    if kv:
        k, v = kv.split("=")
        assert lib.mpmae_set_option(_lib.OPT[k], int(v)) == 0
bf = torch.bfloat16
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, Cc in ((12544, 512), (4864, 320), (77824, 80)):
    x = torch.randn(M, Cc, device=dev).to(bf); xhat = torch.empty_like(x); y = torch.empty_like(x); dx = torch.empty_like(x)
    dy = torch.randn(M, Cc, device=dev).to(bf)
    rstd = torch.empty(M, device=dev); g = torch.rand(Cc, device=dev) + 0.5; b = torch.randn(Cc, device=dev)
    dg = torch.zeros(Cc, device=dev); db = torch.zeros(Cc, device=dev); ws = torch.empty(8 << 20, device=dev)
    print(f"M={M} C={Cc}: {3 * M * Cc * 2 / 1e6:.1f} MB for one read + two writes")
    print(f"  ln_fwd           {t(lambda: lib.mpmae_ln_fwd(1, P(x), P(xhat), P(rstd), P(y), P(g), P(b), 0, 1e-6, M, Cc, None, st)):7.1f} us")
    print(f"  ln_bwd           {t(lambda: lib.mpmae_ln_bwd(1, P(dy), 1, 1.0, P(xhat), P(rstd), P(g), P(b), 0, P(dx), 0, P(dg), P(db), M, Cc, None, P(ws), ws.numel(), st)):7.1f} us")
    print(f"  torch copy (1r+1w) {t(lambda: y.copy_(x)):7.1f} us")
M, D, L = 12544, 512, 49
inv = torch.full((M,), -1, dtype=torch.int32, device=dev)
inv[torch.randperm(M, device=dev)[:4864]] = 1
dxdec = torch.randn(M, D, device=dev).to(bf); dtok = torch.zeros(D, device=dev)
print(f"mask_token_bwd     {t(lambda: lib.mpmae_mask_token_bwd(1, P(dxdec), P(inv), P(dtok), M, D, None, 0, 0, st)):7.1f} us")
