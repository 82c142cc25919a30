"""Stream-K NT kernel (csrc/gemm_sk.cuh, MPMAE_OPT_SK) against the whole-tile kernels (gemm_nt_bf16 / gemm_nt4 / gemm_nt5) and, where the
bundled hipBLASLt can be forced (BLASLT=11), the vendor route: correctness vs torch fp32, stand-alone timing (HIP events, 40 reps)."""
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


ws = torch.empty(256 * 128 * 256, dtype=torch.float32, device=dev)
flags = torch.zeros(L.SK_FLAGS, dtype=torch.int32, device=dev)
for M, N, K, resid, bias_on, act_on in [(12544, 512, 2048, True, True, False), (12544, 512, 2048, False, False, False), (12544, 512, 2816, False, False, False),
                                        (4864, 320, 1280, True, True, True), (4864, 320, 1280, False, False, True), (12544, 2048, 512, False, True, False),
                                        (12544, 2816, 512, False, True, False), (19456, 160, 640, True, True, True)]:
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(bf)
    bias = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev).to(bf)
    act = (torch.rand(M, device=dev) > 0.1).to(torch.uint8)
    g = L.GemmArgs()
    g.A, g.B = a.data_ptr(), w.data_ptr()
    g.bias = bias.data_ptr() if bias_on else 0
    g.act = act.data_ptr() if act_on else 0
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.rpg = M, N, K, K, K, N, M
    g.ws, g.ws_floats, g.sk_flags = ws.data_ptr(), ws.numel(), flags.data_ptr()
    if resid: g.R, g.ldr = r.data_ptr(), N
    ref = a.float() @ w.float().t() + (bias if bias_on else 0) + (r.float() if resid else 0)
    if act_on: ref = ref * act.bool()[:, None]
    line = f"M={M} N={N} K={K} resid={int(resid)} bias={int(bias_on)} mask={int(act_on)}:"
    for name, opts in (("stream-K", dict(SK=2, NT5=0, BLASLT=0)), ("whole tiles", dict(SK=0, NT5=0, BLASLT=0)), ("nt5", dict(SK=0, NT5=1, BLASLT=0)),
                       ("vendor", dict(SK=0, NT5=0, BLASLT=11))):
        for k, v in opts.items(): assert lib.mpmae_set_option(L.OPT[k], v) == 0
        c = torch.full((M, N), 7.0, device=dev, dtype=bf); g.C = c.data_ptr()
        e = lib.mpmae_gemm(1, 0, 2 if resid else 0, C.byref(g), st())
        torch.cuda.synchronize()
        if e != 0:
            line += f"  {name} error {e}"
            continue
        rel = ((c.float() - ref).abs().max() / ref.abs().max()).item()
        us = t_us(lambda: lib.mpmae_gemm(1, 0, 2 if resid else 0, C.byref(g), st()))
        line += f"  {name} {us:6.1f} us ({2 * M * N * K / us / 1e9:.2f} PF/s, rel {rel:.1e})"
    lib.mpmae_set_option(L.OPT["SK"], 0); lib.mpmae_set_option(L.OPT["NT5"], 0); lib.mpmae_set_option(L.OPT["BLASLT"], 0)
    print(line, flush=True)
    assert int(flags.abs().sum()) == 0
