"""Stand-alone time of named ops of the recorded backward (HIP events around back-to-back eager calls on one stream).
usage: [MPMAE_ENGINE_OPTS=...] python tools/probes/op_time.py <substring> [<substring> ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmearth_train_amd.config import make_cfg
from mmearth_train_amd.engine import Engine
from mmearth_train_amd.synth import make_inputs, make_state_dict
import ctypes as C
cfg = make_cfg()
eng = Engine(cfg, 256, dtype="bf16")
eng.load_state_dict(make_state_dict(cfg, seed=0))
eng.set_inputs(*make_inputs(cfg, 256, seed=1000))
for _ in range(2):
    eng.forward(); eng.backward(); eng.optimizer_step(1e-4)
torch.cuda.synchronize()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for lst_name, lst in (("fwd", eng.fwd_ops), ("bwd", eng.bwd_ops)):
    for name, fn, args, meta in lst:
        if not any(t in name for t in sys.argv[1:]):
            continue
        for _ in range(5):
            fn(*args, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn(*args, st)
        e1.record()
        torch.cuda.synchronize()
        print(f"{lst_name} {name:50s} {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us")
