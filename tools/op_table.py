#!/usr/bin/env python
"""Stand-alone (eager, one stream, HIP events around every C-ABI call) time of every op of the step next to its own roofline
(max(algorithmic bytes / 8 TB/s, flops / 2.5 PF)), in program order, with per-lane totals: where the dependent chain is far from
its bound. An op's time includes the second-stage folds its entry point launches.

    MPMAE_ENGINE_OPTS="..." python tools/op_table.py [--batch 256] [--reps 5] [--dtype bf16|fp8] [--subset all_mod|pix_mod]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmearth_train_amd.config import make_cfg  # noqa: E402
from mmearth_train_amd.engine import Engine  # noqa: E402
from mmearth_train_amd.synth import make_inputs, make_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--subset", default="all_mod")
    a = ap.parse_args()
    from mmearth_train_amd import MODALITIES as M
    cfg = make_cfg(out_modalities=M.subset(a.subset))
    eng = Engine(cfg, a.batch, dtype=a.dtype, device="cuda:0")
    eng.load_state_dict(make_state_dict(cfg, seed=0))
    eng.set_inputs(*make_inputs(cfg, a.batch, seed=1))
    eng.forward(); eng.backward()
    torch.cuda.synchronize()
    acc = bench.per_kernel_times(eng, reps=a.reps, by_name=True)
    print(os.environ.get("MPMAE_ENGINE_OPTS", "(default options)"))
    tot = {}
    for phase, ops in (("fwd", eng.fwd_ops), ("bwd", eng.bwd_ops)):
        for name, _, _, m in ops:
            d = acc[name]
            us = d["ms"] / d["n"] * 1e3
            by, fl = d["bytes"] / d["n"], d["flops"] / d["n"]
            roof = max(by / 8e12, fl / 2.5e15) * 1e6
            t = tot.setdefault((phase, m["lane"]), [0.0, 0.0, 0])
            t[0] += us; t[1] += roof; t[2] += 1
            print(f"{phase} lane {m['lane']} {us:8.1f} us  roofline {roof:6.1f} us  x{us / roof if roof > 0 else 0:5.1f}  {by / 1e6:8.1f} MB {fl / 1e9:7.1f} GF  {name}")
    for (phase, lane), (us, roof, n) in sorted(tot.items()):
        print(f"TOTAL {phase} lane {lane}: {n} ops, {us:8.1f} us stand-alone, roofline {roof:7.1f} us")


if __name__ == "__main__":
    main()
