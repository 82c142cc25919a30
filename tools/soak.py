#!/usr/bin/env python
"""Soak run of the fused step (bs 256, bf16, program driver): N optimizer steps on synthetic tiles with fresh mask noise every step;
checks every 250 steps that the loss is finite and decreasing on average, that no update was skipped, and that the persistent stage
kernels never tripped their grid-barrier timeout (ps_sync error words).

    python tools/soak.py [--steps 3000]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd import dist as mdist  # noqa: E402
from mmearth_train_amd.config import make_cfg  # noqa: E402
from mmearth_train_amd.engine import Engine  # noqa: E402
from mmearth_train_amd.synth import make_inputs, make_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    a = ap.parse_args()
    cfg = make_cfg()
    eng = Engine(cfg, 256, dtype="bf16", device="cuda:0")
    eng.load_state_dict(make_state_dict(cfg, seed=0))
    inputs, noise = make_inputs(cfg, 256, seed=1)
    eng.set_inputs(inputs, noise)
    run = mdist.StepRunner(eng, world_size=1, lr=2e-4, mode="program")
    g = torch.Generator(device="cuda:0").manual_seed(7)
    t0 = time.time()
    first = None
    for it in range(a.steps):
        with eng.input_stage(run):
            eng.noise.copy_(torch.randn(eng.noise.shape, device="cuda:0", generator=g))     # a new mask every step
        run.step()
        if (it + 1) % 250 == 0:
            torch.cuda.synchronize()
            m = eng.read_meters()
            loss = m["loss"]["avg"]
            first = first if first is not None else loss
            err = int(eng.ps_sync[:, 2].sum()) if hasattr(eng, "ps_sync") else 0
            print(f"step {it + 1:5d}: loss (window avg) {loss:8.4f}  grad_norm {m['grad_norm']['avg']:8.4f}  skipped {run.skipped_steps()}  "
                  f"ps barrier errors {err}  {(it + 1) / (time.time() - t0):6.1f} steps/s", flush=True)
            assert loss == loss and abs(loss) < 1e6 and run.skipped_steps() == 0 and err == 0
    assert m["loss"]["avg"] < first, "the loss must go down on a fixed batch"
    print("SOAK OK")


if __name__ == "__main__":
    main()
