"""Functional check of the world_size > 1 step driver on ONE GPU: two ranks share cuda:0 and exchange
gradients over gloo (RCCL refuses two ranks per device). Verifies that the segmented launch program +
bucketed all-reduce + AdamW leave both ranks with identical parameters, equal to a single-process run
that averages the two ranks' gradients by hand (the averaged gradients agree to fp32 summation order; the
parameters after two AdamW steps only to ~1e-4, because early-step AdamW divides by |g| and flips on
near-zero gradient entries)."""
import os, sys, torch
import torch.distributed as dist
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


BACKEND = os.environ.get("DDP_PROBE_BACKEND", "gloo")       # "nccl": one rank per GPU over RCCL (needs >= 2 visible GPUs)


def build(rank, N=8, dev=None):
    from mmearth_train_amd.config import make_cfg
    from mmearth_train_amd.engine import Engine
    from mmearth_train_amd.synth import make_inputs, make_state_dict
    cfg = make_cfg()
    eng = Engine(cfg, N, dtype="f32", device=dev or "cuda:0", block_mode="mat")
    eng.load_state_dict(make_state_dict(cfg, seed=0))
    eng.set_inputs(*make_inputs(cfg, N, seed=100 + rank))
    return eng


UF = int(os.environ.get("DDP_PROBE_UPDATE_FREQ", "1"))      # > 1: gradient accumulation over UF micro-steps per update


def worker(rank, world, mode, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29533"
    dev = f"cuda:{rank}" if BACKEND == "nccl" else "cuda:0"
    torch.cuda.set_device(dev)
    if BACKEND == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmearth_train_amd import dist as mdist
    eng = build(rank, dev=dev)
    drv, _, ov = mode.partition("+")              # "program" (bucket events), "program+segments" (one replay call per bucket), "eager"
    run = mdist.StepRunner(eng, world_size=world, lr=1e-3, mode=drv, update_freq=UF, overlap=ov or "events")
    assert drv != "program" or bool(run.bucket_signals) == (ov != "segments")
    for _ in range(UF):
        run.step()
    torch.cuda.synchronize()
    torch.save(eng.gflat.cpu() / world, f"{out}/g{rank}_{mode}.pt")      # all-reduced (summed) gradients of step 1
    for _ in range(UF):
        run.step()
    torch.cuda.synchronize()
    torch.save(eng.pflat.cpu(), f"{out}/p{rank}_{mode}.pt")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
    for mode in ("program", "program+segments", "eager"):
        mp.spawn(worker, args=(2, mode, out), nprocs=2, join=True)
    # reference: one process, both ranks' data, gradients averaged by hand
    engs = [build(0), build(1)]
    gref = None
    for t in (1, 2):
        gs = []
        for e in engs:      # UF identical micro-steps of loss / UF accumulate to the plain gradient
            for u in range(UF):
                e.forward(loss_scale=1.0 / UF); e.backward(zero_grad=(u == 0))
            gs.append(e.gflat.clone())
        g = (gs[0] + gs[1]) / 2
        if gref is None:
            gref = g.cpu().clone()
        for e in engs:
            e.gflat.copy_(g); e.optimizer_step(lr=1e-3)
    torch.cuda.synchronize()
    ref = engs[0].pflat.cpu()
    for mode in ("program", "program+segments", "eager"):
        p0, p1 = torch.load(f"{out}/p0_{mode}.pt"), torch.load(f"{out}/p1_{mode}.pt")
        g0 = torch.load(f"{out}/g0_{mode}.pt")
        print(mode, "step-1 averaged gradient vs reference: max rel %.2e" % ((g0 - gref).abs().max() / gref.abs().max()).item())
        print(mode, "ranks equal:", bool(torch.equal(p0, p1)), " max rel diff vs hand-averaged reference: %.2e" %
              ((p0 - ref).abs().max() / ref.abs().max()).item())
