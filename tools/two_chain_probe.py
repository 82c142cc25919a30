"""Upper bound of a two-chain (half-batch per lane pair) step: two independent engines of batch B/2 stepping concurrently on their own
streams against one engine of batch B. The two half engines do NOT share GRN statistics (so the numbers are a timing bound, not a design)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd.config import make_cfg
from mmearth_train_amd.engine import Engine
from mmearth_train_amd.synth import make_inputs, make_state_dict
from mmearth_train_amd import dist as mdist

dev = torch.device("cuda:0")
cfg = make_cfg()
B = int(os.environ.get("B", "256"))
parts = int(os.environ.get("PARTS", "2"))


def build(bs, seed):
    eng = Engine(cfg, bs, dtype="bf16", device=dev)
    eng.load_state_dict(make_state_dict(cfg, seed=0))
    inputs, noise = make_inputs(cfg, bs, seed=seed)
    eng.set_inputs(inputs, noise)
    return eng, mdist.StepRunner(eng, world_size=1, lr=1e-4, mode="program")


def timed(fn, n=30, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


e1, r1 = build(B, 1)
print(f"one engine, batch {B}: {timed(r1.step):.3f} ms/step", flush=True)
halves = [build(B // parts, 2 + i) for i in range(parts)]
streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]


def both():
    for (e, r), s in zip(halves, streams):
        with torch.cuda.stream(s):
            r.step()


print(f"{parts} engines of batch {B // parts}, concurrent: {timed(both):.3f} ms per step of {B}", flush=True)
print(f"one engine of batch {B // parts} alone: {timed(halves[0][1].step):.3f} ms", flush=True)
