#!/usr/bin/env python
"""RCCL executed on ONE GPU: a process group of one rank on the "nccl" backend (= RCCL on ROCm) drives StepRunner through the
bucketed exchange (four buckets on the communication stream, keyed to the launch program's "bucket ready" events:
mpmae_program_export_signal / mpmae_program_stream_wait + dist.all_reduce(async_op=True), plus the scalar loss all-reduce), and
the result is compared with the plain single-GPU step on the same weights and batch. Reference: DistributedDataParallel
(/root/reference/main_pretrain.py:306-310), init_distributed_mode (/root/reference/helpers.py:337-401).

    python tools/rccl_world1_probe.py          # prints "... max rel ..." lines, exit code 0 on agreement
"""
import os
import socket
import sys

import torch
import torch.distributed as tdist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd import dist as mdist  # noqa: E402
from mmearth_train_amd.config import make_cfg  # noqa: E402
from mmearth_train_amd.engine import Engine  # noqa: E402
from mmearth_train_amd.synth import make_inputs, make_state_dict  # noqa: E402


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


def main():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    mdist.init(backend="nccl", local_rank=0)
    assert tdist.get_backend() == "nccl" and tdist.get_world_size() == 1
    cfg = make_cfg()
    N = 4
    sd = make_state_dict(cfg, seed=5)
    inputs, noise = make_inputs(cfg, N, seed=6)
    ok = True
    for mode in ("program", "eager"):
        res = {}
        for exch in (False, True):
            eng = Engine(cfg, N, dtype="f32", device="cuda:0")
            eng.load_state_dict(sd)
            eng.set_inputs(inputs, noise)
            run = mdist.StepRunner(eng, world_size=1, lr=1e-3, mode=mode, force_exchange=exch)
            assert run.exchange == exch and (len(run.buckets) == 4) == exch
            run.measure_comm_tail = exch                 # (backward end, collectives done) event pairs on the main stream
            for _ in range(2):
                run.step()
            torch.cuda.synchronize()
            if exch:
                tail = run.comm_tail_ms()
                assert tail is not None and tail >= 0.0, tail
                print(f"{mode}: exposed communication tail {tail:.4f} ms per step", flush=True)
            res[exch] = (eng.gflat.clone(), eng.pflat.clone(), run.mean_loss())
        g, p = rel(res[True][0], res[False][0]), rel(res[True][1], res[False][1])
        dl = abs(res[True][2] - res[False][2]) / abs(res[False][2])
        print(f"{mode}: RCCL world-1 exchange vs no exchange after 2 steps: gradients max rel {g:.3e}, parameters max rel {p:.3e}, "
              f"logged loss rel {dl:.3e}", flush=True)
        ok = ok and g < 1e-5 and p < 1e-5 and dl < 1e-5
    mdist.barrier()
    mdist.shutdown()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
