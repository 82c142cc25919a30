"""Yardstick, not product code: the vendor BLAS (torch.nn.functional.linear / torch.mm -> hipBLASLt / rocBLAS) on the step's GEMM shapes, bf16,
stand-alone, next to this library's kernels on the same shapes (profiles/r04/op_table_standalone.txt). Answers one question: is the ~0.45 PF/s
of the decoder / head GEMMs a property of the machine (per-CU operand fill) or of these kernels?"""
import torch
import torch.nn.functional as F

dev = "cuda:0"
bf = torch.bfloat16


def t_us(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [("decoder pw1 / pw2.dgrad  NT", 12544, 2048, 512), ("decoder pw2 / pw1.dgrad  NT", 12544, 512, 2048),
          ("pixel heads              NT", 12544, 2816, 512), ("heads dgrad              NT", 12544, 512, 2816),
          ("stage 2 pw1              NT", 14336, 640, 160), ("stage 2 pw2              NT", 14336, 160, 640),
          ("stage 3 pw1              NT", 3584, 1280, 320), ("stage 3 pw2              NT", 3584, 320, 1280),
          ("stage 0 pw1              NT", 311296, 160, 40), ("stage 1 pw1              NT", 77824, 320, 80)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=bf); w = torch.randn(N, K, device=dev, dtype=bf)
    us = t_us(lambda: F.linear(a, w))
    print(f"{name}  M={M:6d} N={N:5d} K={K:5d}: {us:7.1f} us  {2 * M * N * K / us / 1e9:7.3f} PF/s")
print("weight gradients (TN: dW[N][K] = dY[M][N]^T X[M][K])")
for name, M, N, K in [("decoder pw1.wgrad", 12544, 2048, 512), ("decoder pw2.wgrad", 12544, 512, 2048), ("heads wgrad", 12544, 2816, 512),
                      ("stage 2 pw1.wgrad", 14336, 640, 160), ("stage 0 pw1.wgrad", 311296, 160, 40), ("stage 1 pw1.wgrad", 77824, 320, 80)]:
    dy = torch.randn(M, N, device=dev, dtype=bf); x = torch.randn(M, K, device=dev, dtype=bf)
    us = t_us(lambda: torch.mm(dy.t(), x))
    print(f"{name:24s}  M={M:6d} N={N:5d} K={K:5d}: {us:7.1f} us  {2 * M * N * K / us / 1e9:7.3f} PF/s")
