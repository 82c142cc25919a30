"""Developer check + timing of mpmae_wgrad (bf16 TN weight gradient) against torch fp32 at the step's shapes."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd import _lib
lib = _lib.load()
dev, bf = 'cuda', torch.bfloat16
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ws = torch.empty(32 << 20, dtype=torch.float32, device=dev)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [(311296, 40, 160), (311296, 160, 40), (77824, 80, 320), (77824, 320, 80), (19456, 160, 640), (19456, 640, 160),
          (4864, 320, 1280), (4864, 1280, 320), (12544, 512, 2048), (12544, 2048, 512), (1000, 40, 160), (33, 160, 40)]
tot = 0.0
for M, N, K in shapes:
    torch.manual_seed(M + N)
    Pm = torch.randn(M, N, device=dev).to(bf); Qm = torch.randn(M, K, device=dev).to(bf)
    dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    a = _lib.WgradArgs()
    a.P, a.Q, a.M, a.Nn, a.Kk, a.ldp, a.ldq = Pm.data_ptr(), Qm.data_ptr(), M, N, K, N, K
    a.dW, a.sn, a.sk, a.db = dW.data_ptr(), K, 1, db.data_ptr()
    a.ws, a.ws_floats = ws.data_ptr(), ws.numel()
    err = lib.mpmae_wgrad(1, 0, 0, C.byref(a), 64, st)
    torch.cuda.synchronize()
    assert err == 0, err
    ref = Pm.float().t() @ Qm.float(); rb = Pm.float().sum(0)
    e1 = ((dW - ref).abs().max() / ref.abs().max()).item(); e2 = ((db - rb).abs().max() / rb.abs().max()).item()
    us = timeit(lambda: lib.mpmae_wgrad(1, 0, 0, C.byref(a), 64, st))
    byt = M * (N + K) * 2
    if M > 2000: tot += us
    print(f"M={M:7d} N={N:5d} K={K:5d}  err dW {e1:.2e} db {e2:.2e}  {us:8.1f} us  {byt / us / 1e6:7.2f} TB/s  {2 * M * N * K / us / 1e6:7.1f} TF/s")
print(f"sum over step shapes: {tot:.1f} us")
