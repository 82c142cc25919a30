#!/usr/bin/env python
"""Timing-only decomposition of the backward piece (numerically meaningless: ops are dropped): HIP-event time of (a) the whole
backward, (b) its main-lane ops alone (the data-gradient chain with the GPU to itself), (c) its side-lane ops alone, back to back on
one stream (the weight-gradient work with the GPU to itself). (a) >= max(b, c); (a) close to b + c means the lanes do not overlap.

    MPMAE_ENGINE_OPTS="..." python tools/lane_split_time.py [--batch 256] [--reps 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd.config import make_cfg  # noqa: E402
from mmearth_train_amd.engine import Engine  # noqa: E402
from mmearth_train_amd.synth import make_inputs, make_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    cfg = make_cfg()
    eng = Engine(cfg, a.batch, dtype="bf16", device="cuda:0")
    eng.load_state_dict(make_state_dict(cfg, seed=0))
    eng.set_inputs(*make_inputs(cfg, a.batch, seed=1))
    pieces = eng.step_pieces()
    fwd, zero, bwd, opt = pieces
    main_only = [op for op in bwd if op[3]["lane"] == 0]
    main_only = [(n, f, ar, dict(m, wait=(), signal=None)) for n, f, ar, m in main_only]
    side_only = [(n, f, ar, dict(m, lane=0, wait=(), signal=None)) for n, f, ar, m in bwd if m["lane"] != 0]
    zero0 = [(n, f, ar, dict(m, lane=0, wait=(), signal=None)) for n, f, ar, m in zero]
    fwd_main = [(n, f, ar, dict(m, wait=(), signal=None)) for n, f, ar, m in fwd if m["lane"] == 0]
    prog, spans = eng.record_program([fwd, zero, bwd, main_only, side_only, zero0, fwd_main])
    eng.run_program(prog, (0, spans[2][0] + spans[2][1]))
    torch.cuda.synchronize()

    def timed(span):
        for _ in range(3):
            eng.run_program(prog, span)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            eng.run_program(prog, span)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.reps

    print(os.environ.get("MPMAE_ENGINE_OPTS", "(default options)"))
    print(f"forward piece                 {timed(spans[0]):8.1f} us ({spans[0][1]} ops)")
    print(f"forward, main-lane ops alone  {timed(spans[6]):8.1f} us ({spans[6][1]} ops; results invalid)")
    print(f"backward, both lanes          {timed(spans[2]):8.1f} us ({spans[2][1]} ops)")
    print(f"backward, main-lane ops alone {timed(spans[3]):8.1f} us ({spans[3][1]} ops; results invalid)")
    print(f"backward, side-lane ops alone {timed(spans[4]):8.1f} us ({spans[4][1]} ops, one stream)")


if __name__ == "__main__":
    main()
