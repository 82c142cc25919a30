"""Deep-K NT kernel (csrc/gemm_nt5.cuh, MPMAE_OPT_NT5) against the vendor route and the 128 x 128 kernel: correctness vs torch fp32, timing."""
import ctypes as C, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmearth_train_amd import _lib as L
lib = L.load()
bf, dev = torch.bfloat16, "cuda:0"
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def t_us(fn, n=40):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K, resid, bias_on, act_on in [(12544, 512, 2048, True, True, False), (12544, 512, 2816, False, False, False), (12544, 512, 2048, False, False, True),
                                        (4100, 392, 1408, True, True, True), (3584, 320, 1280, True, True, True), (12544, 2048, 1024, False, True, False)]:
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(bf)
    bias = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev).to(bf)
    act = (torch.rand(M, device=dev) > 0.1).to(torch.uint8)
    g = L.GemmArgs()
    g.A, g.B = a.data_ptr(), w.data_ptr()
    g.bias = bias.data_ptr() if bias_on else 0
    g.act = act.data_ptr() if act_on else 0
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.rpg = M, N, K, K, K, N, M
    if resid: g.R, g.ldr = r.data_ptr(), N
    ref = a.float() @ w.float().t() + (bias if bias_on else 0) + (r.float() if resid else 0)
    if act_on: ref = ref * act.bool()[:, None]
    line = f"M={M} N={N} K={K} resid={int(resid)} bias={int(bias_on)} mask={int(act_on)}:"
    for name, opts in (("nt5", dict(NT5=1)), ("vendor", dict(NT5=0, BLASLT=1)), ("128x128", dict(NT5=0, BLASLT=0))):
        for k, v in opts.items(): assert lib.mpmae_set_option(L.OPT[k], v) == 0
        c = torch.full((M, N), 7.0, device=dev, dtype=bf); g.C = c.data_ptr()
        assert lib.mpmae_gemm(1, 0, 2 if resid else 0, C.byref(g), st()) == 0
        torch.cuda.synchronize()
        rel = ((c.float() - ref).abs().max() / ref.abs().max()).item()
        us = t_us(lambda: lib.mpmae_gemm(1, 0, 2 if resid else 0, C.byref(g), st()))
        line += f"  {name} {us:6.1f} us ({2 * M * N * K / us / 1e9:.2f} PF/s, rel {rel:.1e}{'' if not act_on else ', masked rows zero ' + str(bool((c[~act.bool()] == 0).all()))})"
    lib.mpmae_set_option(L.OPT["NT5"], 0); lib.mpmae_set_option(L.OPT["BLASLT"], 1)
    print(line)
