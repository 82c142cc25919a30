"""Developer check + timing of the fused row-streaming kernels (mpmae_rs) against torch fp32 on the same bf16 values."""
import ctypes as C, sys, os, math, torch
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


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def gelu(x): return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))
def dgelu(x): return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


def args(M, Cc, H, **kw):
    a = _lib.RsArgs()
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
    a.M, a.C, a.H, a.ws, a.ws_floats = M, Cc, H, ws.data_ptr(), ws.numel()
    return a


def run(M, Cc, timing):
    H = 4 * Cc
    torch.manual_seed(M + Cc)
    act = (torch.rand(M, device=dev) > 0.1).to(torch.uint8)
    live = act.bool()[:, None]
    out = {}
    # ---- which 0: LN + pw1 + GELU^2 sums
    d = (torch.randn(M, Cc, device=dev) * 2 + 0.3).to(bf) * live
    lnw = torch.rand(Cc, device=dev) + 0.5; lnb = torch.randn(Cc, device=dev) * 0.1
    W1 = (torch.randn(H, Cc, device=dev) / math.sqrt(Cc)).to(bf); b1 = torch.randn(H, device=dev) * 0.1
    xhat = torch.empty(M, Cc, device=dev, dtype=bf); xn = torch.empty_like(xhat); rstd = torch.empty(M, device=dev)
    h = torch.empty(M, H, device=dev, dtype=bf); s0 = torch.zeros(H, device=dev)
    a0 = args(M, Cc, H, A=d, W=W1, ldw=Cc, bias=b1, v0=lnw, v1=lnb, out=h, xhat=xhat, xn=xn, rstd=rstd, act=act, s0=s0)
    assert lib.mpmae_rs(0, C.byref(a0), st) == 0
    df = d.float(); mu = df.mean(1, keepdim=True); var = ((df - mu) ** 2).mean(1, keepdim=True)
    r_rstd = torch.rsqrt(var + 1e-6)
    r_xhat = (((df - mu) * r_rstd) * live).to(bf)
    r_xn = ((r_xhat.float() * lnw + lnb) * live).to(bf)
    r_h = ((r_xn.float() @ W1.float().t() + b1) * live).to(bf)
    out["0:xhat"] = rel(xhat, r_xhat); out["0:xn"] = rel(xn, r_xn); out["0:rstd"] = rel(rstd, (r_rstd[:, 0] * live[:, 0]))
    out["0:h"] = rel(h, r_h); out["0:s0"] = rel(s0, (gelu(h.float()) ** 2).sum(0))
    # ---- which 4: GRN apply + pw2 + residual
    scale = torch.rand(H, device=dev) + 0.5; gbeta = torch.randn(H, device=dev) * 0.1
    W2 = (torch.randn(Cc, H, device=dev) / math.sqrt(H)).to(bf); b2 = torch.randn(Cc, device=dev) * 0.1
    x = torch.randn(M, Cc, device=dev).to(bf) * live
    z = torch.empty(M, H, device=dev, dtype=bf); o = torch.empty(M, Cc, device=dev, dtype=bf)
    a4 = args(M, Cc, H, A=h, W=W2, ldw=H, bias=b2, v0=scale, v1=gbeta, out=o, xn=z, R=x, act=act, rpg=0)
    assert lib.mpmae_rs(4, C.byref(a4), st) == 0
    r_z = ((gelu(h.float()) * scale + gbeta) * live).to(bf)
    r_o = ((x.float() + r_z.float() @ W2.float().t() + b2) * live).to(bf)
    out["4:z"] = rel(z, r_z); out["4:out"] = rel(o, r_o)
    # ---- which 1: pw2.dgrad + stats
    dout = (torch.randn(M, Cc, device=dev) * 0.1).to(bf) * live
    W2T = W2.t().contiguous()
    dz = torch.empty(M, H, device=dev, dtype=bf); t0 = torch.zeros(H, device=dev); t1 = torch.zeros(H, device=dev)
    a1 = args(M, Cc, H, A=dout, W=W2T, ldw=Cc, out=dz, R=h, s0=t0, s1=t1)
    assert lib.mpmae_rs(1, C.byref(a1), st) == 0
    r_dz = (dout.float() @ W2T.float().t()).to(bf)
    out["1:dz"] = rel(dz, r_dz); out["1:s0"] = rel(t0, dz.float().sum(0)); out["1:s1"] = rel(t1, (dz.float() * gelu(h.float())).sum(0))
    # ---- which 5: dh + pw1.dgrad + LN bwd
    coef = torch.randn(H, device=dev) * 0.05
    W1T = W1.t().contiguous()
    dzc = dz.clone(); dd = torch.empty(M, Cc, device=dev, dtype=bf)
    dg = torch.zeros(Cc, device=dev); db = torch.zeros(Cc, device=dev)
    gbuf = torch.zeros(2 * Cc, device=dev)          # dgamma, dbeta in one buffer (s1 - s0 must be an int offset)
    a5 = args(M, Cc, H, A=dzc, A2=h, W=W1T, ldw=H, v0=scale, v1=coef, out=dd, xhat=xhat, rstd=rstd, lng=lnw, act=act,
              s0=gbuf, s1=gbuf[Cc:], rpg=0)
    assert lib.mpmae_rs(5, C.byref(a5), st) == 0
    hf = h.float()
    r_dh = ((dz.float() * scale + coef * gelu(hf)) * dgelu(hf)).to(bf)
    r_dxn = ((r_dh.float() @ W1T.float().t()).to(bf).float()) * live
    xh = xhat.float(); gq = r_dxn * lnw
    r_dd = ((rstd[:, None] * (gq - gq.mean(1, keepdim=True) - xh * (gq * xh).mean(1, keepdim=True))) * live).to(bf)
    out["5:dh"] = rel(dzc, r_dh); out["5:dd"] = rel(dd, r_dd)
    out["5:dgamma"] = rel(gbuf[:Cc], (r_dxn * xh).sum(0)); out["5:dbeta"] = rel(gbuf[Cc:], r_dxn.sum(0))
    extra = []
    if Cc <= 80:        # dz never materialised: which 1 statistics only, which 5 recomputes dz = dout W2
        u0 = torch.zeros(H, device=dev); u1 = torch.zeros(H, device=dev)
        a1n = args(M, Cc, H, A=dout, W=W2T, ldw=Cc, out=None, R=h, s0=u0, s1=u1)
        assert lib.mpmae_rs(1, C.byref(a1n), st) == 0
        dhb = torch.empty(M, H, device=dev, dtype=bf); dd2 = torch.empty(M, Cc, device=dev, dtype=bf)
        gbuf2 = torch.zeros(2 * Cc, device=dev)
        a5n = args(M, Cc, H, A=dhb, A2=h, W=W1T, ldw=H, v0=scale, v1=coef, out=dd2, xhat=xhat, rstd=rstd, lng=lnw, act=act,
                   s0=gbuf2, s1=gbuf2[Cc:], rpg=0, dz_dout=dout, dz_w2t=W2T, dz_ldw2=Cc)
        assert lib.mpmae_rs(5, C.byref(a5n), st) == 0
        torch.cuda.synchronize()
        out["1n:s0"] = rel(u0, t0); out["1n:s1"] = rel(u1, t1)
        out["5n:dh-5:dh"] = rel(dhb, dzc); out["5n:dd-5:dd"] = rel(dd2, dd)
        out["5n:ndiff"] = float((dhb != dzc).sum().item())
        # h never materialised: which 0 statistics only, which 4 recomputes h = xn W1^T + b1
        v0_ = torch.zeros(H, device=dev)
        xh2 = torch.empty_like(xhat); xn2 = torch.empty_like(xn); rs2 = torch.empty_like(rstd)
        a0n = args(M, Cc, H, A=d, W=W1, ldw=Cc, bias=b1, v0=lnw, v1=lnb, out=None, xhat=xh2, xn=xn2, rstd=rs2, act=act, s0=v0_)
        assert lib.mpmae_rs(0, C.byref(a0n), st) == 0
        z2 = torch.empty_like(z); o2 = torch.empty_like(o)
        a4n = args(M, Cc, H, A=h, W=W2, ldw=H, bias=b2, v0=scale, v1=gbeta, out=o2, xn=z2, R=x, act=act, rpg=0,
                   dz_dout=xn, dz_w2t=W1, dz_ldw2=Cc, dz_bias=b1)
        assert lib.mpmae_rs(4, C.byref(a4n), st) == 0
        torch.cuda.synchronize()
        out["0n:s0"] = rel(v0_, s0); out["4n:z-4:z"] = rel(z2, z); out["4n:out-4:out"] = rel(o2, o)
        extra = [("0 (no h store)", 0, a0n), ("4 (h recomputed)", 4, a4n), ("1 (no dz store)", 1, a1n), ("5 (dz recomputed)", 5, a5n)]
    torch.cuda.synchronize()
    msg = "  ".join(f"{k} {v:.1e}" for k, v in out.items())
    print(f"M={M} C={Cc}: {msg}")
    if timing:
        for w, a in ((0, a0), (4, a4), (1, a1), (5, a5)):
            print(f"    which {w}: {timeit(lambda: lib.mpmae_rs(w, C.byref(a), st)):7.1f} us")
        for nm_, w, a in extra:
            print(f"    which {nm_}: {timeit(lambda: lib.mpmae_rs(w, C.byref(a), st)):7.1f} us")


cases = [(1000, 160, False), (76, 160, False), (19456, 160, True), (304, 320, False), (4864, 320, True),
         (1000, 40, False), (311296, 40, True), (77, 80, False), (77824, 80, True)]
if os.environ.get("RS_ONLY"):
    cases = [c for c in cases if c[2] and c[1] == int(os.environ["RS_ONLY"])]
for M, Cc, tm in cases:
    run(M, Cc, tm)
