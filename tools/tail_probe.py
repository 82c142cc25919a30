"""How long does the main lane wait for the weight-gradient lane at the end of the backward program?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd.config import make_cfg
from mmearth_train_amd.engine import Engine
from mmearth_train_amd.synth import make_inputs, make_state_dict
import mmearth_train_amd.engine as E
cfg = make_cfg()
eng = Engine(cfg, 256, dtype="bf16", device="cuda:0")
eng.load_state_dict(make_state_dict(cfg, seed=0))
eng.set_inputs(*make_inputs(cfg, 256, seed=1))
main = torch.cuda.current_stream()
orig_wait = torch.cuda.Stream.wait_stream
marks = []
def wait_stream(self, other):
    if self.cuda_stream == main.cuda_stream and other.cuda_stream != main.cuda_stream:      # the join at the end of _run
        e = torch.cuda.Event(enable_timing=True); e.record(main); marks.append(("main_done", e))
        orig_wait(self, other)
        e2 = torch.cuda.Event(enable_timing=True); e2.record(main); marks.append(("joined", e2))
    else:
        orig_wait(self, other)
torch.cuda.Stream.wait_stream = wait_stream
for it in range(6):
    marks.clear()
    e0 = torch.cuda.Event(enable_timing=True); e0.record(main)
    eng.forward()
    ef = torch.cuda.Event(enable_timing=True); ef.record(main)
    eng.backward()
    torch.cuda.synchronize()
    md = [e for n, e in marks if n == "main_done"][-1]; jn = [e for n, e in marks if n == "joined"][-1]
    print(f"forward {e0.elapsed_time(ef):.2f} ms, backward main lane {ef.elapsed_time(md):.2f} ms, wait for side lane {md.elapsed_time(jn):.3f} ms")
