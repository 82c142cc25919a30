#!/usr/bin/env python
"""What does the data-parallel step driver cost on ONE GPU? StepRunner with the bucketed RCCL exchange forced on (world size 1: the
collectives move nothing, but every host call, stream wait, bucket event and the separate optimizer replay of the N > 1 path run)
against the plain single-GPU step, bs 256 bf16. The difference is the fixed overhead every rank of an N-GPU run pays on top of the
wire time of its all-reduces.

    python tools/exchange_cost_probe.py [--steps 40]
"""
import argparse
import os
import socket
import sys
import time

import torch
import torch.distributed as tdist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd import dist as mdist  # noqa: E402
from mmearth_train_amd.config import make_cfg  # noqa: E402
from mmearth_train_amd.engine import Engine  # noqa: E402
from mmearth_train_amd.synth import make_inputs, make_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    mdist.init(backend="nccl", local_rank=0)
    cfg = make_cfg()
    sd = make_state_dict(cfg, seed=0)
    inputs, noise = make_inputs(cfg, 256, seed=1)
    for label, kw in (("plain single-GPU step", dict()), ("exchange forced (events)", dict(force_exchange=True)),
                      ("plain single-GPU step", dict()), ("exchange forced (events)", dict(force_exchange=True)),
                      ("exchange forced, bf16 wire", dict(force_exchange=True, allreduce_dtype=torch.bfloat16)),
                      ("exchange forced (segments)", dict(force_exchange=True, overlap="segments"))):
        eng = Engine(cfg, 256, dtype="bf16", device="cuda:0")
        eng.load_state_dict(sd)
        eng.set_inputs(inputs, noise)
        run = mdist.StepRunner(eng, world_size=1, lr=1e-4, mode="program", **kw)
        for _ in range(8):
            run.step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(a.steps):
            run.step()
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"{label:32s} {e0.elapsed_time(e1) / a.steps:7.3f} ms/step (host enqueue {1e3 * (t1 - t0) / a.steps:6.3f} ms/step)", flush=True)
    mdist.barrier()
    mdist.shutdown()


if __name__ == "__main__":
    main()
