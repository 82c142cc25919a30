"""Developer timing of the TN weight-gradient GEMM (mpmae_wgrad, bf16) at the shapes of the bench step, alone on the GPU:
dW[Nn][Kk] = P^T Q over M rows. Prints the transpose-read kernel + second-stage reduce together (what a step pays)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd import _lib
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [(12544, 2816, 512, "head pix"), (12544, 2048, 512, "dec pw1"), (12544, 512, 2048, "dec pw2"), (12544, 512, 320, "proj"),
          (4864, 1280, 320, "s3 pw1"), (4864, 320, 1280, "s3 pw2"), (19456, 640, 160, "s2 pw1"), (19456, 160, 640, "s2 pw2"),
          (77824, 320, 80, "s1 pw1"), (77824, 80, 320, "s1 pw2"), (311296, 160, 40, "s0 pw1"), (311296, 40, 160, "s0 pw2")]
for kv in os.environ.get("LIBOPTS", "").split(","):
    if kv:
        k, v = kv.split("=")
        assert lib.mpmae_set_option(_lib.OPT[k], int(v)) == 0
only = os.environ.get("ONLY")
if only:
    shapes = [x for x in shapes if x[3] == only]
ws = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
for M, N, K, name in shapes:
    torch.manual_seed(1)
    Pm = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    Qm = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    dW = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    a = _lib.WgradArgs()
    a.P, a.Q, a.M, a.Nn, a.Kk, a.ldp, a.ldq = Pm.data_ptr(), Qm.data_ptr(), M, N, K, N, K
    a.dW, a.sn, a.sk, a.db = dW.data_ptr(), K, 1, db.data_ptr()
    a.ws, a.ws_floats = ws.data_ptr(), ws.numel()
    assert lib.mpmae_wgrad(1, 0, 0, C.byref(a), 16, st) == 0
    torch.cuda.synchronize()
    ref = Pm.float().t() @ Qm.float()
    err = ((dW - ref).abs().max() / ref.abs().max()).item()
    us = t(lambda: lib.mpmae_wgrad(1, 0, 0, C.byref(a), 16, st))
    usv = t(lambda: torch.matmul(Pm.t(), Qm))
    mb = (M * (N + K) * 2) / 1e6
    print(f"{name:9s} M={M:6d} Nn={N:4d} Kk={K:4d}: {us:6.1f} us {2*M*N*K/us/1e6:6.0f} TF  {mb/us*1e-3*1e3:6.0f} GB/s operands | vendor {usv:6.1f} us | err {err:.1e}")
