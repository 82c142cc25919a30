"""Developer timing / check of the NT GEMM paths at the decoder, head and wide-stage shapes: the bf16 kernels (mpmae_gemm), the vendor
GEMM through torch, and the MX-fp8 variant (mpmae_quant_mx + mpmae_gemm_mx); statistics epilogues against fp32 column sums."""
import ctypes as C, math, sys, os, torch
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


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


shapes = [(12544, 2048, 512, "dec pw1"), (12544, 512, 2048, "dec pw2"), (12544, 2816, 512, "head pix fwd"),
          (12544, 512, 2816, "head pix dgrad"), (77824, 768, 192, "tiny s1 pw1"), (77824, 192, 768, "tiny s1 pw2"),
          (19456, 1536, 384, "tiny s2 pw1"), (19456, 384, 1536, "tiny s2 pw2"), (4864, 3072, 768, "tiny s3 pw1"), (4864, 768, 3072, "tiny s3 pw2"),
          (4000, 1280, 320, "ragged M"),
          # the short-grid products of the atto step (<= 1 tile per CU: stage 3, proj, downsample convolutions and their data gradients)
          (4864, 320, 1280, "s3 pw2/pw1.dg"), (4864, 1280, 320, "s3 pw1 (plain)"), (4864, 512, 320, "proj"), (4864, 320, 512, "proj.dgrad"),
          (77824, 80, 160, "ds0 conv"), (19456, 160, 320, "ds1 conv"), (4864, 320, 640, "ds2 conv"), (19456, 320, 160, "ds1 dgrad"), (4864, 640, 320, "ds2 dgrad"),
          (4864, 768, 3072, "tiny s3 pw2"), (4864, 512, 768, "tiny proj")]
if os.environ.get("ONLY"):
    shapes = [s_ for s_ in shapes if any(k in s_[3] for k in os.environ["ONLY"].split(","))]
ws = torch.empty(16 << 20, device="cuda")
RINGS = [int(v) for v in os.environ.get("RINGS", "").split(",") if v]      # NT_RING values to time next to the default kernel (e.g. RINGS=332,432,364)
for M, N, K, name in shapes:
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", dtype=bf); w = torch.randn(N, K, device="cuda", dtype=bf) / K ** 0.5
    bias = torch.randn(N, device="cuda"); c = torch.empty(M, N, device="cuda", dtype=bf)
    r = torch.randn(M, N, device="cuda", dtype=bf)
    act = (torch.rand(M, device="cuda") > 0.05).to(torch.uint8)
    g = _lib.GemmArgs()
    g.A, g.B, g.bias, g.C = a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.rpg = M, N, K, K, K, N, M
    g.ws, g.ws_floats = ws.data_ptr(), ws.numel()
    ref = torch.nn.functional.linear(a.float(), w.float(), bias)
    assert lib.mpmae_gemm(1, 0, 0, C.byref(g), st) == 0
    err0 = rel(c.float(), ref)
    us = t(lambda: lib.mpmae_gemm(1, 0, 0, C.byref(g), st))
    us0 = us
    usv = t(lambda: torch.nn.functional.linear(a, w))
    ring_txt = ""
    for rv in RINGS:
        lib.mpmae_set_option(_lib.OPT["NT_RING"], rv)
        try:
            c.zero_()
            assert lib.mpmae_gemm(1, 0, 0, C.byref(g), st) == 0
            er = rel(c.float(), ref)
            ur = t(lambda: lib.mpmae_gemm(1, 0, 0, C.byref(g), st))
        finally:
            lib.mpmae_set_option(_lib.OPT["NT_RING"], 0)
        ring_txt += f" | ring {rv}: {ur:6.1f} us {2*M*N*K/ur/1e6:5.0f} TF err {er:.0e}"
    print(f"{name:15s} M={M} N={N} K={K}: bf16 {us:6.1f} us {2*M*N*K/us/1e6:6.0f} TF (128-row tiles {us0:6.1f} us) | vendor {usv:6.1f} us {2*M*N*K/usv/1e6:6.0f} TF | rel err {err0:.1e}{ring_txt}", flush=True)
    if os.environ.get("BRIEF"):
        continue
    # epilogues: residual + row mask, GELU^2 sums, dz statistics
    g.R, g.ldr, g.act = r.data_ptr(), N, act.data_ptr()
    assert lib.mpmae_gemm(1, 0, 2, C.byref(g), st) == 0
    live = act.bool()[:, None]
    e_res = rel(c.float(), (ref + r.float()) * live)
    s0, s1 = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    g.s0, g.s1 = s0.data_ptr(), s1.data_ptr()
    assert lib.mpmae_gemm(1, 0, 1, C.byref(g), st) == 0          # GELU_SUMSQ
    hq = (ref * live).to(bf).float()
    e_g2 = rel(s0, (gelu(hq) ** 2).sum(0))
    s0.zero_(); g.act = 0
    assert lib.mpmae_gemm(1, 0, 3, C.byref(g), st) == 0          # DZ_STATS: R = h
    dzq = ref.to(bf).float()
    e_s0, e_s1 = rel(s0, dzq.sum(0)), rel(s1, (dzq * gelu(r.float())).sum(0))
    print(f"{'':15s} epilogues: resid {e_res:.1e}  gelu^2 sums {e_g2:.1e}  dz sums {e_s0:.1e} {e_s1:.1e}")
    if K % 128 == 0:
        qa, qw = torch.empty(M, K, dtype=torch.uint8, device="cuda"), torch.empty(N, K, dtype=torch.uint8, device="cuda")
        sa, sw = torch.zeros(K // 128, M, dtype=torch.int32, device="cuda"), torch.zeros(K // 128, N, dtype=torch.int32, device="cuda")
        assert lib.mpmae_quant_mx(a.data_ptr(), K, M, K, qa.data_ptr(), sa.data_ptr(), M, st) == 0
        assert lib.mpmae_quant_mx(w.data_ptr(), K, N, K, qw.data_ptr(), sw.data_ptr(), N, st) == 0
        g8 = _lib.GemmArgs()
        g8.A, g8.B, g8.bias, g8.C = qa.data_ptr(), qw.data_ptr(), bias.data_ptr(), c.data_ptr()
        g8.M, g8.N, g8.K, g8.lda, g8.ldb, g8.ldc, g8.rpg = M, N, K, K, K, N, M
        assert lib.mpmae_gemm_mx(0, C.byref(g8), sa.data_ptr(), M, sw.data_ptr(), N, st) == 0
        torch.cuda.synchronize()
        # dequantised operands on the host side of the check: exact products of what the kernel multiplies
        def deq(q, s, rows):
            e = q.view(torch.float8_e4m3fn).float().view(rows, K // 32, 32)
            sc = s.view(K // 128, rows).t().reshape(-1).clone().view(torch.uint8).view(rows, K // 128, 4).reshape(rows, K // 32).float() - 127
            return (e * torch.exp2(sc)[:, :, None]).view(rows, K)
        da, dw = deq(qa, sa, M), deq(qw, sw, N)
        ref8 = da @ dw.t() + bias
        us8 = t(lambda: lib.mpmae_gemm_mx(0, C.byref(g8), sa.data_ptr(), M, sw.data_ptr(), N, st))
        usq = t(lambda: lib.mpmae_quant_mx(a.data_ptr(), K, M, K, qa.data_ptr(), sa.data_ptr(), M, st))
        print(f"{'':15s} mx-fp8: {us8:6.1f} us {2*M*N*K/us8/1e6:6.0f} TF (+ quant A {usq:5.1f} us) | vs dequantised fp32 product {rel(c.float(), ref8):.1e}"
              f" | quantisation error of A {rel(da, a.float()):.1e}, end to end vs bf16 product {rel(c.float(), ref):.1e}")
