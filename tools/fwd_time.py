#!/usr/bin/env python
"""Unprofiled wall time of the pieces of the recorded step program (HIP events around back-to-back replays of a program range):
forward piece alone, forward + backward, whole step - for A/B of engine options without the profiler's host overhead.

    MPMAE_ENGINE_OPTS="stem_front=0" python tools/fwd_time.py [--batch 256] [--reps 50]
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
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    cfg = make_cfg()
    eng = Engine(cfg, a.batch, dtype="bf16", device="cuda:0")
    eng.load_state_dict(make_state_dict(cfg, seed=0))
    eng.set_inputs(*make_inputs(cfg, a.batch, seed=1))
    run = mdist.StepRunner(eng, world_size=1, lr=1e-4, mode="program")
    for _ in range(5):
        run.step()
    torch.cuda.synchronize()
    nseg = len(run.segments)
    FWD, ZERO, SEG0, OPT = 0, 1, 2, 2 + nseg

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.reps

    out = {}
    out["forward piece"] = timed(lambda: eng.run_program(run.prog, run._span(FWD, FWD)))
    out["forward + zero + backward"] = timed(lambda: eng.run_program(run.prog, run._span(FWD, OPT - 1)))
    out["backward pieces only"] = timed(lambda: eng.run_program(run.prog, run._span(ZERO, OPT - 1)))
    out["whole step (runner.step)"] = timed(run.step)
    print(os.environ.get("MPMAE_ENGINE_OPTS", "(default options)"), " | ".join(f"{k}: {v:7.1f} us" for k, v in out.items()), flush=True)


if __name__ == "__main__":
    main()
