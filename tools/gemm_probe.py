"""Developer timing of mpmae_gemm (bf16 NT fast path) at the dense decoder / head shapes, beside the vendor GEMM through torch."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd import _lib
lib = _lib.load()
bf = torch.bfloat16
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [(12544, 2048, 512, "dec pw1"), (12544, 512, 2048, "dec pw2"), (12544, 2048, 512, "dec pw2.dgrad"),
          (12544, 512, 2048, "dec pw1.dgrad"), (12544, 512, 2000, "head pix dgrad"), (12544, 2000, 512, "head pix fwd"),
          (4864, 1280, 320, "s3 pw1"), (4864, 320, 1280, "s3 pw2"), (12544, 2816, 512, "head pix fwd (real)"),
          (12544, 512, 2816, "head pix dgrad (real)"), (20480, 160, 320, "down1 conv"), (81920, 80, 160, "down0 conv"), (5120, 320, 640, "down2 conv")]
for M, N, K, name in shapes:
    a = torch.randn(M, K, device="cuda", dtype=bf); w = torch.randn(N, K, device="cuda", dtype=bf) / K ** 0.5
    bias = torch.randn(N, device="cuda"); c = torch.empty(M, N, device="cuda", dtype=bf)
    g = _lib.GemmArgs()
    g.A, g.B, g.bias, g.C = a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.rpg = M, N, K, K, K, N, M
    assert lib.mpmae_gemm(1, 0, 0, C.byref(g), st) == 0
    ref = torch.nn.functional.linear(a.float(), w.float(), bias)
    err = ((c.float() - ref).abs().max() / ref.abs().max()).item()
    us = t(lambda: lib.mpmae_gemm(1, 0, 0, C.byref(g), st))
    usv = t(lambda: torch.nn.functional.linear(a, w))
    print(f"{name:16s} M={M} N={N} K={K}: mine {us:7.1f} us {2*M*N*K/us/1e6:6.1f} TF/s | vendor {usv:7.1f} us {2*M*N*K/usv/1e6:6.1f} TF/s | rel err {err:.1e}")
