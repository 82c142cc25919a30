#!/usr/bin/env python
"""Stand-alone time of the weight-gradient-as-statistics pass (csrc/rst.cuh, mpmae_rs which = 6: T = dout^T gelu(h) + db2, incl. its slab fold)
against the two launches it replaces at the stage-0 / stage-1 shapes of the headline workload: the statistics-only pass (mpmae_rs which = 1,
out = NULL, incl. its fold) and pwconv2's weight gradient with the GRN operand prologue (mpmae_wgrad, gemm_tn2 + fold).
    python tools/probes/rst_probe.py
(the sweep over workgroup counts that fixed the launch shapes - RST_WGS, an option until the round-6 prune - is in profiles/r06/rst_probe.txt)"""
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


def timed(fn, reps=40):
    for _ in range(5):
        assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    wgs = [0]
    for M, Cc in ((311296, 40), (77824, 80)):
        H = 4 * Cc
        torch.manual_seed(M + Cc)
        ws = torch.empty(32 << 20, dtype=torch.float32, device=dev)
        live = (torch.rand(M, device=dev) > 0.05)[:, None]
        h = torch.randn(M, H, device=dev).to(bf) * live
        dout = (torch.randn(M, Cc, device=dev) * 0.1).to(bf) * live
        W2T = (torch.randn(H, Cc, device=dev) / math.sqrt(H)).to(bf)
        scale, beta = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1
        T = torch.zeros(Cc * H + Cc, device=dev)
        s0, s1 = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
        dW, db = torch.zeros(Cc, H, device=dev), torch.zeros(Cc, device=dev)

        def rs(which, **kw):
            a = L.RsArgs()
            for k, v in kw.items():
                setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else (0 if v is None else v))
            a.M, a.C, a.H, a.ws, a.ws_floats = M, Cc, H, ws.data_ptr(), ws.numel()
            return lambda: lib.mpmae_rs(which, C.byref(a), st())

        wa = L.WgradArgs()
        for k, v in dict(P=dout, Q=h, M=M, Nn=Cc, Kk=H, ldp=Cc, ldq=H, dW=dW, sn=H, sk=1, db=db, qp0=scale, qp1=beta).items():
            setattr(wa, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
        wa.rpg, wa.ws, wa.ws_floats = M, ws.data_ptr(), ws.numel()
        tiles = ((Cc + 63) // 64) * ((H + 63) // 64)
        splits = max(1, min((768 + tiles - 1) // tiles, (M + 255) // 256))
        nbytes = (M * Cc + M * H) * 2
        t_stats = timed(rs(1, A=dout, W=W2T, ldw=Cc, out=None, R=h, s0=s0, s1=s1))
        t_wg = timed(lambda: lib.mpmae_wgrad(1, L.PRO["NONE"], L.PRO["GRN"], C.byref(wa), splits, st()))
        print(f"M = {M}, C = {Cc}: {nbytes / 1e6:.0f} MB of operands | statistics pass (which 1, out NULL) {t_stats:6.1f} us = {nbytes / t_stats / 1e6:.2f} TB/s | "
              f"pw2.wgrad with GRN prologue (gemm_tn2) {t_wg:6.1f} us", flush=True)
        W2 = W2T.t().contiguous()
        f0, f1, fW, fb = torch.zeros(H, device=dev), torch.zeros(H, device=dev), torch.zeros(Cc, H, device=dev), torch.zeros(Cc, device=dev)
        for nw in (4, 16):
            for w in wgs:
                old = lib.mpmae_get_option(L.OPT["RST_NW"])
                lib.mpmae_set_option(L.OPT["RST_NW"], nw)
                try:
                    t = timed(rs(6, A=dout, R=h, s0=T, s1=T[Cc * H:]))
                    rows = C.c_int(0)
                    tf = timed(rs(6, A=dout, R=h, W=W2, ldw=H, s0=f0, s1=f1, wg_rows=C.addressof(rows)))
                    tw = timed(lambda: lib.mpmae_rs_wgrad_fold(Cc, H, ws.data_ptr(), rows.value, scale.data_ptr(), beta.data_ptr(), fW.data_ptr(), fb.data_ptr(), st()))
                finally:
                    lib.mpmae_set_option(L.OPT["RST_NW"], old)
                print(f"    which 6, RST_NW = {nw:2d}: raw T + reduce_partials {t:6.1f} us = {nbytes / t / 1e6:.2f} TB/s | with S0 / S1 shares + small fold (main lane) {tf:6.1f} us, "
                      f"dW2 / db2 fold of {rows.value} slab rows (weight-gradient lane) {tw:6.1f} us",
                      flush=True)


if __name__ == "__main__":
    main()
