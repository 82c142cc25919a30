import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from mmearth_train_amd.config import make_cfg
from mmearth_train_amd.engine import Engine
from mmearth_train_amd.synth import make_inputs, make_state_dict
def rel(a, b):
    a, b = a.double(), b.double(); return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
cfg = make_cfg()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sd = make_state_dict(cfg, seed=3); inputs, noise = make_inputs(cfg, N, seed=4)
for dtype in ("bf16", "fp8"):
    for sf in (0, 1):
        for ps in (0, 3):
            e = Engine(cfg, N, dtype=dtype, device="cuda:0", options=dict(stem_front=sf, ps=ps))
            e.load_state_dict(sd); e.set_inputs(inputs, noise)
            outs = []
            for r in range(4):
                if r == 2:
                    e.run_segment("encoder"); e.run_segment("decoder")
                elif r == 3:
                    e.run_segment("decoder")
                else:
                    e.forward()
                torch.cuda.synchronize()
                outs.append(({k: v.float().clone() for k, v in e.preds().items()}, e.x0.float().clone(), e.enc_out.float().clone() if hasattr(e, "enc_out") else None))
            for r in (1, 2, 3):
                worst = max((rel(outs[r][0][k], outs[0][0][k]), k) for k in outs[0][0])
                print(dtype, "stem_front", sf, "ps", ps, "run", r, "vs 0: worst pred", worst, "x0", rel(outs[r][1], outs[0][1]))
