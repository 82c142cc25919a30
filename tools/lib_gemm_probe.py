"""What do the vendor GEMMs (through torch) reach on the dense decoder / head shapes? (potential of a library path)"""
import torch
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
bf = torch.bfloat16
for M, N, K, name in [(12544, 2048, 512, "dec pw1"), (12544, 512, 2048, "dec pw2"), (12544, 2048, 512, "dec pw2.dgrad"),
                      (12544, 512, 1998, "head pix dgrad"), (12544, 1998, 512, "head pix fwd"),
                      (4864, 1280, 320, "s3 pw1"), (4864, 320, 1280, "s3 pw2")]:
    a = torch.randn(M, K, device="cuda", dtype=bf); w = torch.randn(N, K, device="cuda", dtype=bf)
    us = t(lambda: torch.nn.functional.linear(a, w))
    print(f"{name:16s} M={M} N={N} K={K}: {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s")
for M, N, K, name in [(12544, 512, 2048, "dec pw2.wgrad"), (12544, 2048, 512, "dec pw1.wgrad")]:
    p = torch.randn(M, N, device="cuda", dtype=bf); q = torch.randn(M, K, device="cuda", dtype=bf)
    us = t(lambda: p.t() @ q)
    print(f"{name:16s} M={M} N={N} K={K}: {us:7.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s")
