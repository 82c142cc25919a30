"""GPU unit tests of individual C-ABI entry points against plain torch fp32 references computed on
the same (bf16-rounded) inputs. The step-level parity tests (test_hip_parity.py) already cover these
kernels through the engine; here each fused kernel is pinned on its own, including ragged row counts,
inactive rows and padded channel counts."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _lib():
    from mmearth_train_amd import _lib
    return _lib, _lib.load()


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def _dgelu(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize("M,Cc", [(1000, 160), (76, 160), (304, 320), (1000, 40), (77, 80), (4864, 80), (500, 192), (304, 384)])
def test_fused_block_kernels_match_torch(M, Cc):
    """mpmae_rs which = 0, 4, 1, 5 (LN+pw1+GELU^2 sums; GRN apply+pw2+residual; pw2.dgrad+sums; dh+pw1.dgrad+LN bwd).
    Tolerance = a couple of bf16 ulps on stored tensors, 1e-3 on fp32 sums (the bf16 path's GELU is a
    2.6e-5-accurate fit, the reference below is the exact erf form)."""
    L, lib = _lib()
    dev, H = "cuda", 4 * Cc
    torch.manual_seed(M + Cc)
    ws = torch.empty(8 << 20, dtype=torch.float32, device=dev)

    def args(**kw):
        a = L.RsArgs()
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
        a.M, a.C, a.H, a.ws, a.ws_floats = M, Cc, H, ws.data_ptr(), ws.numel()
        return a

    act = (torch.rand(M, device=dev) > 0.1).to(torch.uint8)
    live = act.bool()[:, None]
    d = (torch.randn(M, Cc, device=dev) * 2 + 0.3).to(bf) * live
    lnw = torch.rand(Cc, device=dev) + 0.5
    lnb = torch.randn(Cc, device=dev) * 0.1
    W1 = (torch.randn(H, Cc, device=dev) / math.sqrt(Cc)).to(bf)
    b1 = torch.randn(H, device=dev) * 0.1
    xhat = torch.empty(M, Cc, device=dev, dtype=bf)
    xn = torch.empty_like(xhat)
    rstd = torch.empty(M, device=dev)
    h = torch.empty(M, H, device=dev, dtype=bf)
    s0 = torch.zeros(H, device=dev)
    assert lib.mpmae_rs(0, C.byref(args(A=d, W=W1, ldw=Cc, bias=b1, v0=lnw, v1=lnb, out=h, xhat=xhat, xn=xn, rstd=rstd,
                                        act=act, s0=s0)), _st()) == 0
    df = d.float()
    mu = df.mean(1, keepdim=True)
    r_rstd = torch.rsqrt(((df - mu) ** 2).mean(1, keepdim=True) + 1e-6)
    r_xhat = (((df - mu) * r_rstd) * live).to(bf)
    r_xn = ((r_xhat.float() * lnw + lnb) * live).to(bf)
    r_h32 = (xn.float() @ W1.float().t() + b1) * live              # from the kernel's own xn; the GRN sums use the fp32 h
    assert _rel(xhat, r_xhat) < 8e-3 and _rel(xn, r_xn) < 8e-3 and _rel(h, r_h32.to(bf)) < 8e-3
    assert _rel(rstd, r_rstd[:, 0] * live[:, 0]) < 1e-5
    assert _rel(s0, (_gelu(r_h32) ** 2).sum(0)) < 1e-3

    scale = torch.rand(H, device=dev) + 0.5
    gbeta = torch.randn(H, device=dev) * 0.1
    W2 = (torch.randn(Cc, H, device=dev) / math.sqrt(H)).to(bf)
    b2 = torch.randn(Cc, device=dev) * 0.1
    x = torch.randn(M, Cc, device=dev).to(bf) * live
    z = torch.empty(M, H, device=dev, dtype=bf)
    o = torch.empty(M, Cc, device=dev, dtype=bf)
    assert lib.mpmae_rs(4, C.byref(args(A=h, W=W2, ldw=H, bias=b2, v0=scale, v1=gbeta, out=o, xn=z, R=x, act=act, rpg=0)),
                        _st()) == 0
    r_z = ((_gelu(h.float()) * scale + gbeta) * live).to(bf)
    assert _rel(z, r_z) < 8e-3
    assert _rel(o, ((x.float() + z.float() @ W2.float().t() + b2) * live).to(bf)) < 1e-2

    dout = (torch.randn(M, Cc, device=dev) * 0.1).to(bf) * live
    W2T = W2.t().contiguous()
    dz = torch.empty(M, H, device=dev, dtype=bf)
    t0 = torch.zeros(H, device=dev)
    t1 = torch.zeros(H, device=dev)
    assert lib.mpmae_rs(1, C.byref(args(A=dout, W=W2T, ldw=Cc, out=dz, R=h, s0=t0, s1=t1)), _st()) == 0
    r_dz32 = dout.float() @ W2T.float().t()                          # sums are taken before the bf16 store
    assert _rel(dz, r_dz32.to(bf)) < 8e-3
    assert _rel(t0, r_dz32.sum(0)) < 1e-3 and _rel(t1, (r_dz32 * _gelu(h.float())).sum(0)) < 1e-3

    coef = torch.randn(H, device=dev) * 0.05
    W1T = W1.t().contiguous()
    dzc = dz.clone()
    dd = torch.empty(M, Cc, device=dev, dtype=bf)
    gbuf = torch.zeros(2 * Cc, device=dev)
    assert lib.mpmae_rs(5, C.byref(args(A=dzc, A2=h, W=W1T, ldw=H, v0=scale, v1=coef, out=dd, xhat=xhat, rstd=rstd, lng=lnw,
                                        act=act, s0=gbuf, s1=gbuf[Cc:], rpg=0)), _st()) == 0
    hf = h.float()
    r_dh = ((dz.float() * scale + coef * _gelu(hf)) * _dgelu(hf)).to(bf)
    assert _rel(dzc, r_dh) < 1e-2
    dxn = (dzc.float() @ W1T.float().t()).to(bf).float() * live          # from the kernel's own dh: isolates the LN part
    xh = xhat.float()
    gq = dxn * lnw
    r_dd = ((rstd[:, None] * (gq - gq.mean(1, keepdim=True) - xh * (gq * xh).mean(1, keepdim=True))) * live).to(bf)
    assert _rel(dd, r_dd) < 2e-2
    assert _rel(gbuf[:Cc], (dxn * xh).sum(0)) < 5e-3 and _rel(gbuf[Cc:], dxn.sum(0)) < 5e-3


@pytest.mark.parametrize("dt,S", [(0, 8), (1, 8), (1, 16)])
def test_im2col_of_masked_image_matches_unfold(dt, S):
    L, lib = _lib()
    dev, N, Cin, grid = "cuda", 3, 12, 7
    Himg = grid * S
    keep = 19
    torch.manual_seed(S + dt)
    img = torch.randn(N, Cin, Himg, Himg, device=dev)
    noise = torch.rand(N, grid * grid, device=dev)
    mask = torch.empty(N, grid * grid, device=dev)
    vis = torch.empty(N, keep, dtype=torch.int32, device=dev)
    inv = torch.empty(N, grid * grid, dtype=torch.int32, device=dev)
    assert lib.mpmae_mask_gen(noise.data_ptr(), N, grid * grid, keep, mask.data_ptr(), vis.data_ptr(), inv.data_ptr(), _st()) == 0
    ldo = 112
    out = torch.full((N * keep * S * S, ldo), 7.0, device=dev, dtype=torch.float32 if dt == 0 else bf)
    assert lib.mpmae_im2col3(dt, img.data_ptr(), vis.data_ptr(), inv.data_ptr(), out.data_ptr(), ldo, N, keep, grid, S, Cin,
                             Himg, _st()) == 0
    # reference: zero the masked patches, unfold 3x3 with zero padding, pick the visible patches' pixels
    pm = (inv.view(N, 1, grid, grid) >= 0).float().repeat_interleave(S, 2).repeat_interleave(S, 3)
    cols = torch.nn.functional.unfold(img * pm, 3, padding=1).view(N, Cin, 3, 3, Himg, Himg)      # [n, cin, kh, kw, y, x]
    ref = torch.zeros(N * keep * S * S, ldo, device=dev)
    v = vis.long()
    for n in range(N):
        for slot in range(keep):
            py, px = int(v[n, slot]) // grid, int(v[n, slot]) % grid
            blk = cols[n, :, :, :, py * S:(py + 1) * S, px * S:(px + 1) * S]                         # [cin, kh, kw, S, S]
            r0 = (n * keep + slot) * S * S
            ref[r0:r0 + S * S, :9 * Cin] = blk.permute(3, 4, 2, 1, 0).reshape(S * S, 9 * Cin)       # k = (kw*3+kh)*Cin + cin
    assert _rel(out.float(), ref.to(out.dtype).float()) < 1e-6


@pytest.mark.parametrize("dt", [0, 1])
def test_stem_tail_forward_and_backward_match_autograd(dt):
    L, lib = _lib()
    dev, M, Cc = "cuda", 1000, 40
    T = torch.float32 if dt == 0 else bf
    torch.manual_seed(11 + dt)
    act = (torch.rand(M, device=dev) > 0.15).to(torch.uint8)
    live = act.bool()[:, None]
    x = (torch.randn(M, Cc, device=dev) * 1.5).to(T) * live
    p = {k: (torch.rand(Cc, device=dev) + 0.5) if k in ("g1", "w", "g2") else torch.randn(Cc, device=dev) * 0.1
         for k in ("g1", "b1", "w", "wb", "g2", "b2")}
    xhat1, xhat2, out = (torch.empty(M, Cc, device=dev, dtype=T) for _ in range(3))
    rstd1, rstd2 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ws = torch.empty(4 << 20, device=dev)
    a = L.StemTailArgs()
    a.x, a.out, a.xhat1, a.rstd1, a.xhat2, a.rstd2 = (t.data_ptr() for t in (x, out, xhat1, rstd1, xhat2, rstd2))
    for k in p:
        setattr(a, k, p[k].data_ptr())
    a.act_in = a.act_out = act.data_ptr()
    a.M, a.C, a.ws, a.ws_floats = M, Cc, ws.data_ptr(), ws.numel()
    assert lib.mpmae_stem_tail(dt, 0, C.byref(a), _st()) == 0

    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.float().clone().requires_grad_(True)

    def ln(v, g, b):
        mu = v.mean(1, keepdim=True)
        return (v - mu) * torch.rsqrt(((v - mu) ** 2).mean(1, keepdim=True) + 1e-6) * g + b
    a1 = _gelu(ln(xr, q["g1"], q["b1"])) * live
    s0 = (a1 * q["w"] + q["wb"]) * live
    y = ln(s0, q["g2"], q["b2"]) * live
    tol = 2e-5 if dt == 0 else 2e-2
    assert _rel(out.float(), y) < tol
    dy = (torch.randn(M, Cc, device=dev) * 0.1).to(T) * live
    y.backward(dy.float())
    dx = torch.empty(M, Cc, device=dev, dtype=T)
    g = {k: torch.zeros(Cc, device=dev) for k in ("dg1", "db1", "dw", "dwb", "dg2", "db2")}
    a.x, a.out = dy.data_ptr(), dx.data_ptr()
    for k in g:
        setattr(a, k, g[k].data_ptr())
    assert lib.mpmae_stem_tail(dt, 1, C.byref(a), _st()) == 0
    assert _rel(dx.float(), xr.grad * live) < (1e-4 if dt == 0 else 3e-2)
    for k, r in (("dg1", "g1"), ("db1", "b1"), ("dw", "w"), ("dwb", "wb"), ("dg2", "g2"), ("db2", "b2")):
        assert _rel(g[k], q[r].grad) < (1e-4 if dt == 0 else 3e-2), k


def test_grouped_layernorm_feeds_the_downsample_gemm():
    """mpmae_ln_fwd_down writes row (patch, cy, cx) to [parent][(cx&1)*2 + (cy&1)][C]; mpmae_ln_bwd_down reads dy from it."""
    L, lib = _lib()
    dev, NK, S, Cc = "cuda", 6, 4, 40
    M = NK * S * S
    torch.manual_seed(5)
    act = (torch.rand(M, device=dev) > 0.2).to(torch.uint8)
    live = act.bool()[:, None]
    x = torch.randn(M, Cc, device=dev) * live
    g = torch.rand(Cc, device=dev) + 0.5
    b = torch.randn(Cc, device=dev) * 0.1
    xhat = torch.empty_like(x)
    rstd = torch.empty(M, device=dev)
    yg = torch.full((M // 4, 4 * Cc), 9.0, device=dev)
    assert lib.mpmae_ln_fwd_down(0, x.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), yg.data_ptr(), g.data_ptr(), b.data_ptr(),
                                 1e-6, M, Cc, S, act.data_ptr(), _st()) == 0
    mu = x.mean(1, keepdim=True)
    y = ((x - mu) * torch.rsqrt(((x - mu) ** 2).mean(1, keepdim=True) + 1e-6) * g + b) * live
    yv = y.view(NK, S // 2, 2, S // 2, 2, Cc)                     # [nk, iy, a, ix, b, c]
    ref = yv.permute(0, 1, 3, 4, 2, 5).reshape(M // 4, 4 * Cc)    # group index = b*2 + a
    assert _rel(yg, ref) < 1e-5
    dyg = torch.randn(M // 4, 4 * Cc, device=dev)
    dx = torch.empty_like(x)
    dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    ws = torch.empty(1 << 20, device=dev)
    assert lib.mpmae_ln_bwd_down(0, dyg.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), g.data_ptr(), dx.data_ptr(),
                                 dg.data_ptr(), db.data_ptr(), M, Cc, S, act.data_ptr(), ws.data_ptr(), ws.numel(), _st()) == 0
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    mu = xr.mean(1, keepdim=True)
    yr = ((xr - mu) * torch.rsqrt(((xr - mu) ** 2).mean(1, keepdim=True) + 1e-6) * gr + br) * live
    dy = dyg.view(NK, S // 2, S // 2, 2, 2, Cc).permute(0, 1, 4, 2, 3, 5).reshape(M, Cc)       # back to row order
    yr.backward(dy)
    assert _rel(dx, xr.grad * live) < 1e-4 and _rel(dg, gr.grad) < 1e-4 and _rel(db, br.grad) < 1e-4


@pytest.mark.parametrize("M,Cc", [(1000, 40), (333, 80), (1216, 160)])
def test_folded_grn_finalisation_equals_the_separate_launches(M, Cc):
    """mpmae_rs which = 4 / 5 with fin_* set (GRN finalisation recomputed in the kernel prologue) against
    mpmae_grn_fwd_finalize / mpmae_grn_bwd_finalize followed by the same kernel: same arithmetic in the same
    order, so stored tensors must agree bit for bit and the published vectors to fp32 round-off."""
    L, lib = _lib()
    dev, H = "cuda", 4 * Cc
    torch.manual_seed(3 * M + Cc)
    ws = torch.empty(8 << 20, dtype=torch.float32, device=dev)

    def args(**kw):
        a = L.RsArgs()
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
        a.M, a.C, a.H, a.ws, a.ws_floats = M, Cc, H, ws.data_ptr(), ws.numel()
        return a

    P = lambda t: C.c_void_p(t.data_ptr())
    act = (torch.rand(M, device=dev) > 0.1).to(torch.uint8)
    h = torch.randn(M, H, device=dev).to(bf)
    x = torch.randn(M, Cc, device=dev).to(bf)
    W2 = (torch.randn(Cc, H, device=dev) / math.sqrt(H)).to(bf)
    b2 = torch.randn(Cc, device=dev) * 0.1
    G2 = (torch.rand(H, device=dev) + 0.1) * M
    gamma = torch.randn(H, device=dev) * 0.5
    gbeta = torch.randn(H, device=dev) * 0.1
    eps = 1e-6
    # ---- forward: separate finalize, then the kernel on its outputs
    Gx, Ainv, scale = torch.empty(H, device=dev), torch.empty(1, device=dev), torch.empty(H, device=dev)
    assert lib.mpmae_grn_fwd_finalize(P(G2), P(gamma), eps, 1, H, P(Gx), P(Ainv), P(scale), _st()) == 0
    z0, out0 = torch.empty(M, H, device=dev, dtype=bf), torch.empty(M, Cc, device=dev, dtype=bf)
    assert lib.mpmae_rs(4, C.byref(args(A=h, W=W2, ldw=H, bias=b2, v0=scale, v1=gbeta, out=out0, xn=z0, R=x, act=act)), _st()) == 0
    Gx1, Ainv1, scale1 = torch.zeros(H, device=dev), torch.zeros(1, device=dev), torch.zeros(H, device=dev)
    z1, out1 = torch.empty_like(z0), torch.empty_like(out0)
    assert lib.mpmae_rs(4, C.byref(args(A=h, W=W2, ldw=H, bias=b2, v1=gbeta, out=out1, xn=z1, R=x, act=act, fin_sum=G2,
                                        fin_gamma=gamma, fin_gx=Gx1, fin_ainv=Ainv1, fin_out=scale1, fin_eps=eps)), _st()) == 0
    torch.cuda.synchronize()
    assert torch.equal(z0, z1) and torch.equal(out0, out1)
    assert torch.equal(Gx, Gx1) and torch.equal(Ainv, Ainv1) and torch.equal(scale, scale1)
    # ---- backward
    dz = torch.randn(M, H, device=dev).to(bf)
    S0, S1 = torch.randn(H, device=dev) * M ** 0.5, torch.randn(H, device=dev) * M ** 0.5
    W1T = (torch.randn(Cc, H, device=dev) / math.sqrt(H)).to(bf)
    xhat = torch.randn(M, Cc, device=dev).to(bf)
    rstd = torch.rand(M, device=dev) + 0.5
    lng = torch.rand(Cc, device=dev) + 0.5
    coef, dgam, dbet = torch.empty(H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    assert lib.mpmae_grn_bwd_finalize(P(S0), P(S1), P(Gx), P(Ainv), P(gamma), 1, H, P(coef), P(dgam), P(dbet), _st()) == 0
    dza, dd0 = dz.clone(), torch.empty(M, Cc, device=dev, dtype=bf)
    g0, gb0 = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    assert lib.mpmae_rs(5, C.byref(args(A=dza, A2=h, W=W1T, ldw=H, v0=scale, v1=coef, out=dd0, xhat=xhat, rstd=rstd, lng=lng,
                                        act=act, s0=g0, s1=gb0)), _st()) == 0
    dzb, dd1 = dz.clone(), torch.empty_like(dd0)
    g1, gb1 = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    coef1, dgam1, dbet1 = torch.zeros(H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    assert lib.mpmae_rs(5, C.byref(args(A=dzb, A2=h, W=W1T, ldw=H, v0=scale, out=dd1, xhat=xhat, rstd=rstd, lng=lng, act=act,
                                        s0=g1, s1=gb1, fin_sum=S1, fin_sum0=S0, fin_gamma=gamma, fin_gx=Gx, fin_ainv=Ainv,
                                        fin_out=coef1, fin_dgamma=dgam1, fin_dbeta=dbet1)), _st()) == 0
    torch.cuda.synchronize()
    assert torch.equal(dza, dzb) and torch.equal(dd0, dd1)
    assert torch.equal(coef, coef1) and torch.equal(dgam, dgam1) and torch.equal(dbet, dbet1)
    assert _rel(g0, g1) < 1e-6 and _rel(gb0, gb1) < 1e-6


def test_polynomial_gelu_over_every_bf16_input():
    """The bf16 path's transcendental-free GELU (common.cuh gelu2_fwd) through mpmae_grn_apply with scale 1, beta 0,
    on all 65536 bf16 bit patterns: |error| <= 3.1e-5 before the bf16 rounding of the result (tools/gelu_fit.py)."""
    L, lib = _lib()
    dev = "cuda"
    bits = torch.arange(65536, dtype=torch.int32, device=dev).to(torch.int16)
    h = bits.view(bf).reshape(64, 1024).contiguous()
    z = torch.empty_like(h)
    scale, beta = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    assert lib.mpmae_grn_apply(1, P(h), P(z), P(scale), P(beta), 64, 1024, 64, None, _st()) == 0
    torch.cuda.synchronize()
    x = h.double().flatten()
    fin = torch.isfinite(x)
    ref = (0.5 * x * (1 + torch.erf(x / math.sqrt(2))))[fin]
    got = z.double().flatten()[fin]
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    assert (err <= ref.abs() * 2.0 ** -8 + 4e-5).all(), float((err - ref.abs() * 2.0 ** -8).max())


@pytest.mark.parametrize("M,N,K,resid", [(4200, 512, 2048, True), (4100, 384, 512, False), (5000, 256, 64, False)])
def test_direct_to_lds_gemm_matches_torch(M, N, K, resid):
    """mpmae_gemm bf16 NT with K % 64 == 0 and M >= 4096 takes the global_load_lds path (swizzled unpadded LDS rows,
    ragged last row tile clamped): compare with an fp32 matmul of the same bf16 operands."""
    L, lib = _lib()
    dev = "cuda"
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev).to(bf)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(bf)
    bias = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(bf)
    c = torch.empty(M, N, device=dev, dtype=bf)
    g = L.GemmArgs()
    g.A, g.B, g.bias, g.C = a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.rpg = M, N, K, K, K, N, M
    if resid:
        g.R, g.ldr = r.data_ptr(), N
    assert lib.mpmae_gemm(1, 0, 2 if resid else 0, C.byref(g), _st()) == 0
    ref = a.float() @ w.float().t() + bias + (r.float() if resid else 0)
    assert _rel(c, ref) < 6e-3


@pytest.mark.parametrize("M,Cc", [(1000, 40), (333, 80)])
def test_dz_recomputation_matches_the_materialised_path(M, Cc):
    """mpmae_rs which = 1 with out == NULL (statistics only) and which = 5 with dz_dout / dz_w2t (dz = dout W2 recomputed
    chunk by chunk, never read) against the pair that stores and re-reads dz. Same MFMA operands and order, so dz is
    identical; dh / dd may differ by a bf16 ulp where the two instantiations contract an fma differently."""
    L, lib = _lib()
    dev, H = "cuda", 4 * Cc
    torch.manual_seed(7 * M + Cc)
    ws = torch.empty(8 << 20, dtype=torch.float32, device=dev)

    def args(**kw):
        a = L.RsArgs()
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
        a.M, a.C, a.H, a.ws, a.ws_floats = M, Cc, H, ws.data_ptr(), ws.numel()
        return a

    act = (torch.rand(M, device=dev) > 0.1).to(torch.uint8)
    live = act.bool()[:, None]
    h = torch.randn(M, H, device=dev).to(bf)
    dout = (torch.randn(M, Cc, device=dev) * 0.1).to(bf) * live
    W2T = (torch.randn(H, Cc, device=dev) / math.sqrt(H)).to(bf)
    W1T = (torch.randn(Cc, H, device=dev) / math.sqrt(H)).to(bf)
    scale, coef = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.05
    xhat = torch.randn(M, Cc, device=dev).to(bf)
    rstd, lng = torch.rand(M, device=dev) + 0.5, torch.rand(Cc, device=dev) + 0.5
    dz = torch.empty(M, H, device=dev, dtype=bf)
    s0, s1, u0, u1 = (torch.zeros(H, device=dev) for _ in range(4))
    assert lib.mpmae_rs(1, C.byref(args(A=dout, W=W2T, ldw=Cc, out=dz, R=h, s0=s0, s1=s1)), _st()) == 0
    assert lib.mpmae_rs(1, C.byref(args(A=dout, W=W2T, ldw=Cc, out=None, R=h, s0=u0, s1=u1)), _st()) == 0
    dd0, dd1 = torch.empty(M, Cc, device=dev, dtype=bf), torch.empty(M, Cc, device=dev, dtype=bf)
    g0, g1 = torch.zeros(2 * Cc, device=dev), torch.zeros(2 * Cc, device=dev)
    dh0, dh1 = dz.clone(), torch.empty_like(dz)
    assert lib.mpmae_rs(5, C.byref(args(A=dh0, A2=h, W=W1T, ldw=H, v0=scale, v1=coef, out=dd0, xhat=xhat, rstd=rstd, lng=lng,
                                        act=act, s0=g0, s1=g0[Cc:])), _st()) == 0
    assert lib.mpmae_rs(5, C.byref(args(A=dh1, A2=h, W=W1T, ldw=H, v0=scale, v1=coef, out=dd1, xhat=xhat, rstd=rstd, lng=lng,
                                        act=act, s0=g1, s1=g1[Cc:], dz_dout=dout, dz_w2t=W2T, dz_ldw2=Cc)), _st()) == 0
    torch.cuda.synchronize()
    assert _rel(u0, s0) < 1e-5 and _rel(u1, s1) < 1e-5
    assert _rel(dh1, dh0) < 5e-3 and (dh1 != dh0).float().mean().item() < 1e-3
    assert _rel(dd1, dd0) < 1e-2 and _rel(g1, g0) < 1e-3
    # forward twin: which = 4 recomputing h = xn W1^T + b1 is bit-identical to reading the h that which = 0 stored
    d = (torch.randn(M, Cc, device=dev) * 2 + 0.3).to(bf) * live
    lnw, lnb = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
    W1 = (torch.randn(H, Cc, device=dev) / math.sqrt(Cc)).to(bf)
    b1, b2, gbeta = torch.randn(H, device=dev) * 0.1, torch.randn(Cc, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1
    xh, xn, rs_ = torch.empty(M, Cc, device=dev, dtype=bf), torch.empty(M, Cc, device=dev, dtype=bf), torch.empty(M, device=dev)
    hs, q0, q1 = torch.empty(M, H, device=dev, dtype=bf), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    assert lib.mpmae_rs(0, C.byref(args(A=d, W=W1, ldw=Cc, bias=b1, v0=lnw, v1=lnb, out=hs, xhat=xh, xn=xn, rstd=rs_, act=act, s0=q0)), _st()) == 0
    assert lib.mpmae_rs(0, C.byref(args(A=d, W=W1, ldw=Cc, bias=b1, v0=lnw, v1=lnb, out=None, xhat=xh, xn=xn, rstd=rs_, act=act, s0=q1)), _st()) == 0
    x = torch.randn(M, Cc, device=dev).to(bf) * live
    W2 = W2T.t().contiguous()
    za, oa, zb, ob = (torch.empty(M, H, device=dev, dtype=bf), torch.empty(M, Cc, device=dev, dtype=bf),
                      torch.empty(M, H, device=dev, dtype=bf), torch.empty(M, Cc, device=dev, dtype=bf))
    assert lib.mpmae_rs(4, C.byref(args(A=hs, W=W2, ldw=H, bias=b2, v0=scale, v1=gbeta, out=oa, xn=za, R=x, act=act)), _st()) == 0
    assert lib.mpmae_rs(4, C.byref(args(A=hs, W=W2, ldw=H, bias=b2, v0=scale, v1=gbeta, out=ob, xn=zb, R=x, act=act, dz_dout=xn,
                                        dz_w2t=W1, dz_ldw2=Cc, dz_bias=b1)), _st()) == 0
    torch.cuda.synchronize()
    assert _rel(q1, q0) < 1e-5 and torch.equal(za, zb) and torch.equal(oa, ob)
    # and against plain torch on the same bf16 operands
    hf, r_dz = h.float(), (dout.float() @ W2T.float().t()).to(bf)
    r_dh = ((r_dz.float() * scale + coef * _gelu(hf)) * _dgelu(hf)).to(bf)
    assert _rel(dh1, r_dh) < 1.5e-2


@pytest.mark.parametrize("M,Cc", [(1000, 40), (64, 40), (333, 80), (77824, 80), (100000, 40)])
def test_weight_gradient_as_statistics_pass_matches_torch_and_the_statistics_kernel(M, Cc):
    """mpmae_rs which = 6 (csrc/rst.cuh, round 6): T = dout^T gelu(h), db2 = sum_rows dout in ONE read of dout and h, against fp32 matmuls on the
    same bf16 operands (the kernel rounds gelu(h) to bf16 like every MFMA operand of the bf16 mode), accumulating into non-zero outputs; then
    mpmae_grn_stats_from_wgrad on its result against the statistics pass it replaces (mpmae_rs which = 1, out = NULL) and the parameter
    gradients dW2 = dout^T (gelu(h) * scale + beta), db2. Ragged row counts, one tile, more tiles than workgroups, both widths."""
    L, lib = _lib()
    dev, H = "cuda", 4 * Cc
    torch.manual_seed(3 * M + Cc)
    ws = torch.empty(24 << 20, dtype=torch.float32, device=dev)

    def args(**kw):
        a = L.RsArgs()
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
        a.M, a.C, a.H, a.ws, a.ws_floats = M, Cc, H, ws.data_ptr(), ws.numel()
        return a

    live = (torch.rand(M, device=dev) > 0.1)[:, None]
    h = torch.randn(M, H, device=dev).to(bf)
    dout = (torch.randn(M, Cc, device=dev) * 0.1).to(bf) * live
    T = torch.full((Cc * H + Cc,), 0.5, device=dev)                   # T | db2 adjacent, as in the engine's arena; the call ADDS
    assert lib.mpmae_rs(6, C.byref(args(A=dout, R=h, s0=T, s1=T[Cc * H:])), _st()) == 0
    torch.cuda.synchronize()
    g = _gelu(h.float()).to(bf).float()
    Tref = dout.float().t() @ g
    dbref = dout.float().sum(0)
    scl = Tref.abs().max().item()
    assert (T[:Cc * H].view(Cc, H) - 0.5 - Tref).abs().max().item() <= 2e-3 * scl      # (the kernel's GELU is a 3e-5-accurate fit: a bf16 ulp of g now and then)
    assert (T[Cc * H:] - 0.5 - dbref).abs().max().item() <= 1e-4 * dbref.abs().max().item() + 1e-5
    # the statistics and parameter gradients it stands for
    T -= 0.5
    W2 = (torch.randn(Cc, H, device=dev) / math.sqrt(H)).to(bf)          # [C][H]: the staged forward weight
    W2T = W2.t().contiguous()
    scale, beta = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1
    dW2, db2, S0, S1 = torch.zeros(Cc, H, device=dev), torch.zeros(Cc, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    assert lib.mpmae_grn_stats_from_wgrad(1, T.data_ptr(), T[Cc * H:].data_ptr(), W2.data_ptr(), H, scale.data_ptr(), beta.data_ptr(), dW2.data_ptr(),
                                          db2.data_ptr(), S0.data_ptr(), S1.data_ptr(), Cc, H, _st()) == 0
    u0, u1 = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    assert lib.mpmae_rs(1, C.byref(args(A=dout, W=W2T, ldw=Cc, out=None, R=h, s0=u0, s1=u1)), _st()) == 0
    torch.cuda.synchronize()
    # (S1 through T carries one bf16 rounding of gelu(h) per term - the rounding pwconv2's weight gradient has always had - where the statistics
    # kernel multiplies the fp32 value: a relative 2^-9 per term, ~ 1e-3 of a random-sign sum; bound 6e-3 of the largest column)
    assert (S0 - u0).abs().max().item() <= 2e-3 * u0.abs().max().item() + 1e-4
    assert (S1 - u1).abs().max().item() <= 6e-3 * u1.abs().max().item() + 1e-4
    z = (g * scale + beta)
    assert _rel(dW2, dout.float().t() @ z) < 3e-3 and _rel(db2, dbref) < 1e-4
    # the fused form (what the engine issues): every workgroup emits its share of S0 / S1 (folded by the call itself); the big slabs stay in ws for
    # mpmae_rs_wgrad_fold -> dW2, db2 (the weight-gradient lane's op); T is never stored
    f0, f1, fW, fb = torch.ones(H, device=dev), torch.ones(H, device=dev), torch.ones(Cc, H, device=dev), torch.ones(Cc, device=dev)
    rows = C.c_int(0)
    assert lib.mpmae_rs(6, C.byref(args(A=dout, R=h, W=W2, ldw=H, s0=f0, s1=f1, wg_rows=C.addressof(rows))), _st()) == 0
    assert rows.value >= 1
    assert lib.mpmae_rs_wgrad_fold(Cc, H, ws.data_ptr(), rows.value, scale.data_ptr(), beta.data_ptr(), fW.data_ptr(), fb.data_ptr(), _st()) == 0
    torch.cuda.synchronize()
    for got, ref in ((f0 - 1, S0), (f1 - 1, S1), (fW - 1, dW2), (fb - 1, db2)):
        assert (got - ref).abs().max().item() <= 1e-4 * ref.abs().max().item() + 1e-5      # (same products, another summation order)


@pytest.mark.parametrize("M", [1000, 64, 40000])
def test_pointwise1_weight_gradient_inside_the_fused_backward_kernel(M):
    """MpmaeRsArgs.wg_ws (round 6, C = 40): mpmae_rs which = 5 accumulates U = dh^T x-hat and db1 = sum_rows dh per persistent workgroup and does
    NOT store dh; mpmae_rs_wgrad_fold applies the LayerNorm affine by linearity. Against the same call without wg_ws (dd, the LayerNorm gamma / beta
    partials: identical arithmetic) and against fp32 matmuls on the dh that call stored: dW1 = dh^T (x-hat * gamma + beta), db1 = sum dh."""
    L, lib = _lib()
    dev, Cc = "cuda", 40
    H = 4 * Cc
    torch.manual_seed(11 * M)
    ws, ws2 = torch.empty(8 << 20, dtype=torch.float32, device=dev), torch.empty(8 << 20, dtype=torch.float32, device=dev)

    def args(**kw):
        a = L.RsArgs()
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
        a.M, a.C, a.H, a.ws, a.ws_floats = M, Cc, H, ws.data_ptr(), ws.numel()
        return a

    act = (torch.rand(M, device=dev) > 0.1).to(torch.uint8)
    live = act.bool()[:, None]
    h = torch.randn(M, H, device=dev).to(bf) * live
    dout = (torch.randn(M, Cc, device=dev) * 0.1).to(bf) * live
    W2T = (torch.randn(H, Cc, device=dev) / math.sqrt(H)).to(bf)
    W1T = (torch.randn(Cc, H, device=dev) / math.sqrt(H)).to(bf)
    scale, coef = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.05
    xhat = torch.randn(M, Cc, device=dev).to(bf) * live
    rstd, lng, lnb = (torch.rand(M, device=dev) + 0.5) * act, torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.3
    dd0, dd1 = torch.empty(M, Cc, device=dev, dtype=bf), torch.empty(M, Cc, device=dev, dtype=bf)
    g0, g1 = torch.zeros(2 * Cc, device=dev), torch.zeros(2 * Cc, device=dev)
    dh0, dh1 = torch.zeros(M, H, device=dev, dtype=bf), torch.full((M, H), 7.0, device=dev, dtype=bf)
    rows = C.c_int(0)
    common = dict(A2=h, W=W1T, ldw=H, v0=scale, v1=coef, xhat=xhat, rstd=rstd, lng=lng, act=act, dz_dout=dout, dz_w2t=W2T, dz_ldw2=Cc)
    assert lib.mpmae_rs(5, C.byref(args(A=dh0, out=dd0, s0=g0, s1=g0[Cc:], **common)), _st()) == 0
    assert lib.mpmae_rs(5, C.byref(args(A=dh1, out=dd1, s0=g1, s1=g1[Cc:], wg_ws=ws2, wg_ws_floats=ws2.numel(), wg_rows=C.addressof(rows), **common)),
                        _st()) == 0
    dW1, db1 = torch.full((H, Cc), 0.25, device=dev), torch.full((H,), 0.25, device=dev)
    assert rows.value >= 1
    assert lib.mpmae_rs_wgrad_fold(H, Cc, ws2.data_ptr(), rows.value, lng.data_ptr(), lnb.data_ptr(), dW1.data_ptr(), db1.data_ptr(), _st()) == 0
    torch.cuda.synchronize()
    assert torch.equal(dd1, dd0) and _rel(g1, g0) < 1e-5
    assert (dh1 == 7.0).all(), "dh must not be written in the fused form"
    xn = xhat.float() * lng + lnb
    ref_w, ref_b = dh0.float().t() @ xn, dh0.float().sum(0)
    assert _rel(dW1 - 0.25, ref_w) < 2e-4 and _rel(db1 - 0.25, ref_b) < 2e-4


@pytest.mark.parametrize("N,keep,S,Cc", [(3, 19, 8, 40), (2, 19, 4, 80), (1, 5, 4, 96)])
def test_downsample_layernorm_fused_into_the_pointwise2_kernel(N, keep, S, Cc):
    """MpmaeRsArgs.dn_* (round 6): mpmae_rs which = 4 also emits the LayerNorm in front of the 2x2/2 downsample convolution - x-hat, rstd and the affine
    output in the convolution's grouped [M/4][4C] operand layout - from the row it has just computed, against mpmae_ln_fwd_down on the stored bf16 row
    (identical arithmetic on identical bf16 inputs: a bf16 ulp where the two kernels contract an fma differently); inactive rows write zeros; `out`
    may be NULL."""
    L, lib = _lib()
    dev, H = "cuda", 4 * Cc
    M = N * keep * S * S
    torch.manual_seed(N + S + Cc)
    ws = torch.empty(4 << 20, dtype=torch.float32, device=dev)

    def args(**kw):
        a = L.RsArgs()
        for k, v in kw.items():
            setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
        a.M, a.C, a.H, a.ws, a.ws_floats = M, Cc, H, ws.data_ptr(), ws.numel()
        return a

    act = (torch.rand(M, device=dev) > 0.1).to(torch.uint8)
    live = act.bool()[:, None]
    h = torch.randn(M, H, device=dev).to(bf) * live
    x = torch.randn(M, Cc, device=dev).to(bf) * live
    W2 = (torch.randn(Cc, H, device=dev) / math.sqrt(H)).to(bf)
    b2, scale, gbeta = torch.randn(Cc, device=dev) * 0.1, torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1
    lng, lnb = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.2
    out0, xh0, rs0, y0 = (torch.empty(M, Cc, device=dev, dtype=bf), torch.empty(M, Cc, device=dev, dtype=bf), torch.empty(M, device=dev),
                          torch.empty(M // 4, 4 * Cc, device=dev, dtype=bf))
    xh1, rs1, y1 = torch.empty_like(xh0), torch.empty_like(rs0), torch.empty_like(y0)
    assert lib.mpmae_rs(4, C.byref(args(A=h, W=W2, ldw=H, bias=b2, v0=scale, v1=gbeta, out=out0, R=x, act=act)), _st()) == 0
    assert lib.mpmae_ln_fwd_down(1, out0.data_ptr(), xh0.data_ptr(), rs0.data_ptr(), y0.data_ptr(), lng.data_ptr(), lnb.data_ptr(), 1e-6, M, Cc, S,
                                 act.data_ptr(), _st()) == 0
    assert lib.mpmae_rs(4, C.byref(args(A=h, W=W2, ldw=H, bias=b2, v0=scale, v1=gbeta, out=None, R=x, act=act, dn_xhat=xh1, dn_rstd=rs1, dn_y=y1,
                                        dn_gamma=lng, dn_beta=lnb, dn_S=S)), _st()) == 0
    torch.cuda.synchronize()
    assert _rel(rs1, rs0) < 1e-5 and (rs1[~act.bool()] == 0).all()
    assert _rel(xh1, xh0) < 8e-3 and (xh1 != xh0).float().mean().item() < 2e-2
    assert _rel(y1, y0) < 8e-3 and (y1.view(M // 4, 4, Cc)[:, :, :] != 0).any()
    # inactive rows: zeros in both outputs
    assert (xh1[~act.bool()] == 0).all()
