#!/usr/bin/env python
"""A/B of the persistent per-sample stage kernels (csrc/ps.cuh, engine option ps=1) against the row-streaming block kernels
(ps=0) on the same weights, inputs and mask: per-block tensors of stages 2 / 3, losses, gradients, and HIP-event timings.

    python tools/ps_check.py [--batch 256] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd.config import make_cfg  # noqa: E402
from mmearth_train_amd.engine import Engine  # noqa: E402
from mmearth_train_amd.synth import make_inputs, make_state_dict  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def time_ops(eng, ops, reps):
    st = eng._stream()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        for name, fn, args, _m in ops:
            assert fn(*args, st) == 0, name
    e0.record()
    for _ in range(reps):
        for name, fn, args, _m in ops:
            assert fn(*args, st) == 0, name
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--bwd", action="store_true")
    a = ap.parse_args()
    cfg = make_cfg()
    sd = make_state_dict(cfg, seed=3)
    inputs, noise = make_inputs(cfg, a.batch, seed=4)
    engs = {}
    for ps in (0, 1):
        eng = Engine(cfg, a.batch, dtype="bf16", device="cuda:0", options=dict(ps=3 * ps, z_free=0))
        eng.load_state_dict(sd)
        eng.set_inputs(inputs, noise)
        eng.forward()
        eng.backward()
        torch.cuda.synchronize()
        engs[ps] = eng
    e0, e1 = engs[0], engs[1]
    if hasattr(e1, "ps_sync"):
        print("ps sync rows (arrivals, departures, error):", e1.ps_sync[:e1._ps_launches, :4].tolist())
        for r in range(e1._ps_launches):
            st = e1.ps_sync[r, 8:16].tolist()
            if any(st):
                d = [(st[i + 1] - st[i]) & 0xffffffff for i in range(7)]
                if r < 2:
                    print(f"launch {r} (fwd) block-1 phase cycles (workgroup 0): dw {d[0]} | LN {d[1]} | pw1 {d[2]} | stats+barrier {d[3]} | finalize {d[4]} | z {d[5]} | pw2 {d[6]}  total {sum(d)}")
                else:
                    print(f"launch {r} (bwd) block-1 phase cycles (workgroup 0): dz gemm {d[0]} | sums+barrier {d[1]} | finalize {d[2]} | dh {d[3]} | dxn gemm + LN bwd {d[4]} | dw dgrad {d[5]}  total {sum(d[:6])}")
    worst = 0.0
    for b0, b1 in zip(e0.blocks, e1.blocks):
        if b0["stage"] not in (2, 3):
            continue
        errs = {k: rel(b1[k].float(), b0[k].float()) for k in ("dhat", "rstd", "xn", "h", "z", "out", "Gx", "scale")}
        errs["Ainv"] = rel(b1["Ainv"][:1], b0["Ainv"][:1])
        worst = max(worst, max(errs.values()))
        print(b0["prefix"], " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    print("losses ps=0:", [round(v, 5) for v in e0.losses.tolist()])
    print("losses ps=1:", [round(v, 5) for v in e1.losses.tolist()])
    print("total", e0.total.item(), e1.total.item(), "rel", abs(e0.total.item() - e1.total.item()) / abs(e0.total.item()))
    bad = []
    for k in e0.grads:
        g0, g1 = e0.grads[k].double().flatten(), e1.grads[k].double().flatten()
        c = torch.nn.functional.cosine_similarity(g0, g1, dim=0).item()
        r = (g1.norm() / (g0.norm() + 1e-30)).item()
        if c < 0.999 or abs(r - 1) > 2e-2:
            bad.append((k, round(c, 5), round(r, 4)))
    print("gradient tensors with cosine < 0.999 or norm ratio off by > 2 %:", bad if bad else "none")
    gcos = torch.nn.functional.cosine_similarity(e0.gflat.double(), e1.gflat.double(), dim=0).item()
    print("flat gradient cosine ps=1 vs ps=0:", gcos, " worst block-tensor rel err:", worst)
    # timings: the stage ops of the forward
    for ps, eng in engs.items():
        for stage in (2, 3):
            ops = [op for op in eng.fwd_ops if op[0].startswith(f"encoder.stages.{stage}")]
            print(f"ps={ps} stage {stage} forward: {len(ops)} ops, {time_ops(eng, ops, a.reps):8.1f} us")
        ops = [op for op in eng.fwd_ops if op[3]["lane"] == 0]
        print(f"ps={ps} whole forward main lane (single stream, eager): {len(ops)} ops, {time_ops(eng, ops, a.reps):8.1f} us")
        if a.bwd:
            for stage in (2, 3):
                ops = [op for op in eng.bwd_ops if op[0].startswith(f"encoder.stages.{stage}") and op[3]["lane"] == 0]
                print(f"ps={ps} stage {stage} backward main lane: {len(ops)} ops, {time_ops(eng, ops, a.reps):8.1f} us")


if __name__ == "__main__":
    main()
