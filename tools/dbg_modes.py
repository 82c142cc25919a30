import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from golden_cases import CASES, case_cfg, case_data
from mmearth_train_amd import dist as mdist
from mmearth_train_amd.engine import Engine
c = CASES["allmod_atto_56"]; cfg = case_cfg(c); sd, inputs, noise = case_data(c, cfg)
def rel(a, b): return ((a.double()-b.double()).abs().max()/(b.double().abs().max()+1e-30)).item()
out = {}
for m in ("eager", "hipgraph", "program"):
    for steps in (1, 2, 3):
        eng = Engine(cfg, c["N"], dtype="f32", device="cuda:0", block_mode="mat"); eng.load_state_dict(sd); eng.set_inputs(inputs, noise)
        run = mdist.StepRunner(eng, world_size=1, lr=1e-3, mode=m)
        for _ in range(steps): run.step()
        torch.cuda.synchronize()
        out[(m, steps)] = (eng.losses.cpu().clone(), eng.gflat.cpu().clone(), eng.pflat.cpu().clone())
for m in ("hipgraph", "program"):
    for steps in (1, 2, 3):
        print(m, steps, [f"{rel(a, b):.2e}" for a, b in zip(out[("eager", steps)], out[(m, steps)])])
