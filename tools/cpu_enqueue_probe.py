"""How long does the host need to ENQUEUE one step (native launch program vs Python loop)?"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd import MODALITIES as M, dist as mdist
from mmearth_train_amd.config import make_cfg
from mmearth_train_amd.engine import Engine
from mmearth_train_amd.synth import make_inputs, make_state_dict
cfg = make_cfg()
for mode in ("program", "eager"):
    eng = Engine(cfg, 256, dtype="bf16", device="cuda:0")
    eng.load_state_dict(make_state_dict(cfg, seed=0))
    eng.set_inputs(*make_inputs(cfg, 256, seed=1))
    run = mdist.StepRunner(eng, world_size=1, lr=1e-4, mode=mode)
    for _ in range(5): run.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): run.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode:8s}: host enqueue {1e3 * (t1 - t0) / 20:.2f} ms/step, wall {1e3 * (t2 - t0) / 20:.2f} ms/step")
