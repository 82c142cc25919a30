#!/usr/bin/env python
"""(Round 6: the workgroup-count options this probe swept - RSP_WGS, RSP_NWGS, RSC1_WGS, RSC1_CPS - are frozen at the values it found; the sweep below
fails with KeyError on those names and is kept as the record of HOW they were found: profiles/r05/rs1_probe.txt, rsp_*_probe.txt.)
Stand-alone time and agreement of the persistent burst-load NARROW fused pointwise kernels (csrc/rsp.cuh, MPMAE_OPT_RSP) against the
chunk-streaming kernels (rsc.cuh) at the stage-0 / stage-1 shapes of the headline workload: mpmae_rs which = 4 (GRN + pw2 + residual, folded
finalisation) and which = 5 (dz recomputed, dh, pw1.dgrad, LayerNorm backward).   python tools/probes/rsp_narrow_probe.py"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmearth_train_amd import _lib as L  # noqa: E402

lib = L.load()
bf = torch.bfloat16
dev = "cuda"


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(M, Cc, which, opts, zfree, reps=30, dzr=True):
    H = 4 * Cc
    torch.manual_seed(M + Cc + which)
    ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)
    act = (torch.rand(M, device=dev) > 0.05).to(torch.uint8)
    live = act.bool()[:, None]
    h = torch.randn(M, H, device=dev).to(bf) * live
    gamma, gbeta = torch.randn(H, device=dev) * 0.5, torch.randn(H, device=dev) * 0.1
    if which == 4:
        x = torch.randn(M, Cc, device=dev).to(bf) * live
        W2 = (torch.randn(Cc, H, device=dev) / math.sqrt(H)).to(bf)
        b2 = torch.randn(Cc, device=dev) * 0.1
        G2 = (torch.rand(H, device=dev) + 0.1) * M
        Gx, Ainv, scale = torch.zeros(H, device=dev), torch.zeros(1, device=dev), torch.zeros(H, device=dev)
        z, out = torch.zeros(M, H, device=dev, dtype=bf), torch.empty(M, Cc, device=dev, dtype=bf)
        kw = dict(A=h, W=W2, ldw=H, bias=b2, v1=gbeta, out=out, xn=None if zfree else z, R=x, act=act, fin_sum=G2, fin_gamma=gamma,
                  fin_gx=Gx, fin_ainv=Ainv, fin_out=scale, fin_eps=1e-6)
        outs = lambda: [out.clone(), z.clone(), scale.clone(), Gx.clone(), Ainv.clone()]
        zero = lambda: None
    else:
        dout = (torch.randn(M, Cc, device=dev) * 0.1).to(bf) * live
        W2T = (torch.randn(H, Cc, device=dev) / math.sqrt(H)).to(bf)
        W1T = (torch.randn(Cc, H, device=dev) / math.sqrt(H)).to(bf)
        scale = torch.rand(H, device=dev) + 0.5
        S0, S1 = torch.randn(H, device=dev) * M ** 0.5, torch.randn(H, device=dev) * M ** 0.5
        Gx, Ainv = torch.rand(H, device=dev) + 0.5, torch.rand(1, device=dev) + 0.5
        xhat = torch.randn(M, Cc, device=dev).to(bf) * live
        rstd, lng = (torch.rand(M, device=dev) + 0.5) * act, torch.rand(Cc, device=dev) + 0.5
        dh, dd = torch.zeros(M, H, device=dev, dtype=bf), torch.empty(M, Cc, device=dev, dtype=bf)
        g = torch.zeros(2 * Cc, device=dev)
        coef, dgam, dbet = torch.zeros(H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
        dzsrc = (torch.randn(M, H, device=dev) * 0.1).to(bf) * live
        if not dzr:
            dh.copy_(dzsrc)
        kw = dict(A=dh, A2=h, W=W1T, ldw=H, v0=scale, out=dd, xhat=xhat, rstd=rstd, lng=lng, act=act, s0=g, s1=g[Cc:],
                  **(dict(dz_dout=dout, dz_w2t=W2T, dz_ldw2=Cc) if dzr else {}), fin_sum=S1, fin_sum0=S0, fin_gamma=gamma, fin_gx=Gx, fin_ainv=Ainv, fin_out=coef, fin_dgamma=dgam, fin_dbeta=dbet)
        outs = lambda: [dd.clone(), dh.clone(), g.clone(), coef.clone(), dgam.clone(), dbet.clone()]
        zero = lambda: (g.zero_(), dgam.zero_(), dbet.zero_(), None if dzr else dh.copy_(dzsrc))
    a = L.RsArgs()
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
    a.M, a.C, a.H, a.ws, a.ws_floats = M, Cc, H, ws.data_ptr(), ws.numel()
    saved = {}
    for k, v in opts.items():
        saved[k] = lib.mpmae_get_option(L.OPT[k])
        lib.mpmae_set_option(L.OPT[k], v)
    try:
        for _ in range(5):
            assert lib.mpmae_rs(which, C.byref(a), st()) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.mpmae_rs(which, C.byref(a), st())
        e1.record()
        torch.cuda.synchronize()
        zero()
        assert lib.mpmae_rs(which, C.byref(a), st()) == 0
        torch.cuda.synchronize()
    finally:
        for k, v in saved.items():
            lib.mpmae_set_option(L.OPT[k], v)
    return e0.elapsed_time(e1) / reps * 1e3, outs()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "ring":      # the ring-pipelined backward kernel (rsn3.cuh) at the stage-2 shape
        for M in (19456, 1000, 76):
            t0, ref = run(M, 160, 5, dict(RSN3=0), False, dzr=False)
            print(f"M={M} C=160 which=5 (dz materialised): rsc_narrow {t0:6.1f} us (incl. fold)")
            for o in (dict(RSN3=4), dict(RSN3=5)):
                t, got = run(M, 160, 5, o, False, dzr=False)
                errs = []
                for g_, r in zip(got, ref):
                    den = r.float().abs().max().item() + 1e-30
                    errs.append((g_.float() - r.float()).abs().max().item() / den)
                print(f"    {str(o):30s} {t:6.1f} us   max rel diff: " + " ".join(f"{e:.1e}" for e in errs))
        return
    for (M, Cc, zfree) in ((311296, 40, True), (77824, 80, False), (1000, 40, False), (333, 80, True)):
        for which in (4, 5):
            t0, ref = run(M, Cc, which, dict(RSP=0), zfree)
            print(f"M={M} C={Cc} which={which}: chunked {t0:6.1f} us")
            vs = [dict(RSP=1), dict(RSP=1, RSP_NWV=4), dict(RSP=1, RSP_NWV=8)]
            if M > 10000:
                vs += [dict(RSP=1, RSP_NWV=n, RSP_NWGS=w) for n in (4, 8) for w in (256, 512, 768, 1024)]
            for o in vs:
                try:
                    t, got = run(M, Cc, which, o, zfree)
                except AssertionError:
                    print(f"    {o}: launch refused")
                    continue
                errs = []
                for g_, r in zip(got, ref):
                    den = r.float().abs().max().item() + 1e-30
                    errs.append((g_.float() - r.float()).abs().max().item() / den)
                print(f"    {str(o):50s} {t:6.1f} us   max rel diff vs chunked: " + " ".join(f"{e:.1e}" for e in errs))


if __name__ == "__main__":
    main()
