#!/usr/bin/env python
"""(Round 6: the workgroup-count options this probe swept - RSP_WGS, RSP_NWGS, RSC1_WGS, RSC1_CPS - are frozen at the values it found; the sweep below
fails with KeyError on those names and is kept as the record of HOW they were found: profiles/r05/rs1_probe.txt, rsp_*_probe.txt.)
Stand-alone time and agreement of the one-shot wide fused pointwise kernels (csrc/rsc1.cuh, MPMAE_OPT_RSC1) against the chunk-streaming
kernels (rsc.cuh) at the stage-2 / stage-3 shapes of the headline workload: mpmae_rs which = 0 (LN + pw1 + GELU^2 sums) and
which = 1 (pw2.dgrad + statistics).  python tools/probes/rs1_probe.py"""
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


def run(M, Cc, which, opts, reps=60):
    H = 4 * Cc
    torch.manual_seed(M + Cc + which)
    ws = torch.empty(16 << 20, dtype=torch.float32, device=dev)
    act = (torch.rand(M, device=dev) > 0.05).to(torch.uint8)
    live = act.bool()[:, None]
    d = (torch.randn(M, Cc, device=dev) * 2 + 0.3).to(bf) * live
    lnw, lnb = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
    W1 = (torch.randn(H, Cc, device=dev) / math.sqrt(Cc)).to(bf)
    b1 = torch.randn(H, device=dev) * 0.1
    h = torch.empty(M, H, device=dev, dtype=bf)
    xhat, xn = torch.empty(M, Cc, device=dev, dtype=bf), torch.empty(M, Cc, device=dev, dtype=bf)
    rstd = torch.empty(M, device=dev)
    s0, s1 = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    hh = (torch.randn(M, H, device=dev)).to(bf)
    a = L.RsArgs()
    kw = (dict(A=d, W=W1, ldw=Cc, bias=b1, v0=lnw, v1=lnb, out=h, xhat=xhat, xn=xn, rstd=rstd, act=act, s0=s0) if which == 0 else
          dict(A=d, W=W1, ldw=Cc, out=h, R=hh, s0=s0, s1=s1))
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    a.M, a.C, a.H, a.ws, a.ws_floats = M, Cc, H, ws.data_ptr(), ws.numel()
    saved = {}
    for k, v in opts.items():
        saved[k] = lib.mpmae_get_option(L.OPT[k])
        lib.mpmae_set_option(L.OPT[k], v)
    try:
        for _ in range(10):
            s0.zero_(); s1.zero_()
            assert lib.mpmae_rs(which, C.byref(a), st()) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.mpmae_rs(which, C.byref(a), st())
        e1.record()
        torch.cuda.synchronize()
        s0.zero_(); s1.zero_()
        assert lib.mpmae_rs(which, C.byref(a), st()) == 0
        torch.cuda.synchronize()
    finally:
        for k, v in saved.items():
            lib.mpmae_set_option(L.OPT[k], v)
    outs = [h.clone(), s0.clone(), s1.clone()] + ([xhat.clone(), xn.clone(), rstd.clone()] if which == 0 else [])
    return e0.elapsed_time(e1) / reps * 1e3, outs


def main():
    small = len(sys.argv) > 1 and sys.argv[1] == "small"
    if small:      # the persistent burst-load kernels (rsp.cuh) at the stage-0 / stage-1 shapes
        for (M, Cc) in ((311296, 40), (77824, 80)):
            for which in (0, 1):
                t0, ref = run(M, Cc, which, dict(RSP=0), reps=30)
                print(f"M={M} C={Cc} which={which}: chunked {t0:6.1f} us (incl. fold)")
                for o in (dict(RSP=1), dict(RSP=2), dict(RSP=1, RSP_WGS=512), dict(RSP=1, RSP_WGS=768), dict(RSP=1, RSP_WGS=1536), dict(RSP=1, RSP_WGS=2048),
                          dict(RSP=2, RSP_WGS=512), dict(RSP=2, RSP_WGS=768), dict(RSP=2, RSP_WGS=1536)):
                    t, got = run(M, Cc, which, o, reps=30)
                    errs = []
                    for g, r in zip(got, ref):
                        den = r.float().abs().max().item() + 1e-30
                        errs.append((g.float() - r.float()).abs().max().item() / den)
                    print(f"    {str(o):44s} {t:6.1f} us   max rel diff vs chunked: " + " ".join(f"{e:.1e}" for e in errs))
        return
    for (M, Cc) in ((19456, 160), (4864, 320)):
        for which in (0, 1):
            t0, ref = run(M, Cc, which, dict(RSC1=0))
            print(f"M={M} C={Cc} which={which}: chunked {t0:6.1f} us (incl. fold)")
            variants = [dict(RSC1=1), dict(RSC1=2), dict(RSC1=1, RSC1_WGS=512), dict(RSC1=2, RSC1_WGS=512), dict(RSC1=1, RSC1_WGS=1536), dict(RSC1=2, RSC1_WGS=1536)]
            if Cc == 160:
                variants += [dict(RSC1=1, RSC1_CPS=64), dict(RSC1=2, RSC1_CPS=64), dict(RSC1=2, RSC1_CPS=64, RSC1_WGS=1536)]
            variants += [dict(RSC1=1, RSC_ATOMIC=400), dict(RSC1=2, RSC_ATOMIC=400)]
            for o in variants:
                try:
                    t, got = run(M, Cc, which, o)
                except AssertionError:
                    print(f"    {o}: launch refused")
                    continue
                errs = []
                for g, r in zip(got, ref):
                    den = r.float().abs().max().item() + 1e-30
                    errs.append((g.float() - r.float()).abs().max().item() / den)
                print(f"    {str(o):44s} {t:6.1f} us   max rel diff vs chunked: " + " ".join(f"{e:.1e}" for e in errs))


if __name__ == "__main__":
    main()
