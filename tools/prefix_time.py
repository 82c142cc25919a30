#!/usr/bin/env python
"""In-situ marginal cost of every op of the forward piece, unprofiled: T(k) = HIP-event time of replaying program ops [0, k)
back to back; the marginal T(k) - T(k-1) is what op k-1 adds to the forward with both lanes live (a side-lane op that hides under
the main lane adds ~0, a main-lane op that waits for the side lane adds its wait).

    MPMAE_ENGINE_OPTS="..." python tools/prefix_time.py [--batch 256] [--reps 30] [--piece 0]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd import dist as mdist  # noqa: E402
from mmearth_train_amd.config import make_cfg  # noqa: E402
from mmearth_train_amd.engine import Engine  # noqa: E402
from mmearth_train_amd.synth import make_inputs, make_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--piece", type=int, default=0)
    a = ap.parse_args()
    cfg = make_cfg()
    eng = Engine(cfg, a.batch, dtype="bf16", device="cuda:0")
    eng.load_state_dict(make_state_dict(cfg, seed=0))
    eng.set_inputs(*make_inputs(cfg, a.batch, seed=1))
    pieces = eng.step_pieces()
    prog, spans = eng.record_program(pieces)
    for _ in range(3):
        eng.run_program(prog, (0, sum(n for _, n in spans) - 2))       # everything but the optimizer
    torch.cuda.synchronize()
    lo, n = spans[a.piece]
    ops = pieces[a.piece]

    def timed(k):
        for _ in range(3):
            eng.run_program(prog, (lo, k))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            eng.run_program(prog, (lo, k))
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.reps

    prev = 0.0
    print(os.environ.get("MPMAE_ENGINE_OPTS", "(default options)"))
    for k in range(1, n + 1):
        t = timed(k)
        name, _, _, m = ops[k - 1]
        print(f"{k:3d} lane {m['lane']} {t:8.1f} us  +{t - prev:7.1f}  {name}  wait={list(m['wait'])} signal={m['signal']}", flush=True)
        prev = t


if __name__ == "__main__":
    main()
