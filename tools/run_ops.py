"""Developer tool: run the engine's programs a few times eagerly (for rocprofv3 --pmc runs)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd import MODALITIES as M
from mmearth_train_amd.config import make_cfg
from mmearth_train_amd.engine import Engine
from mmearth_train_amd.synth import make_inputs, make_state_dict
cfg = make_cfg()
eng = Engine(cfg, 256, dtype="bf16")
eng.load_state_dict(make_state_dict(cfg, seed=0))
inputs, noise = make_inputs(cfg, 256, seed=1000)
eng.set_inputs(inputs, noise)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    eng.forward(); eng.backward(); eng.optimizer_step(1e-4)
torch.cuda.synchronize()
print("done", eng.total.item())
