"""Run-to-run spread of the forward: the same engine, weights, inputs and mask noise, forward 6 times; per block output and the losses
against the first run, in bf16 ulps of the tensor's max (<= 2 bf16 ulps at stages 2-3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmearth_train_amd.config import make_cfg
from mmearth_train_amd.engine import Engine
from mmearth_train_amd.synth import make_inputs, make_state_dict
N = int(os.environ.get("N", "256"))
cfg = make_cfg()
e = Engine(cfg, N, dtype="bf16", device="cuda:0")
e.load_state_dict(make_state_dict(cfg, seed=3))
inputs, noise = make_inputs(cfg, N, seed=4)
e.set_inputs(inputs, noise)
runs = []
for r in range(6):
    e.forward()
    torch.cuda.synchronize()
    runs.append(([b["out"].float().clone() for b in e.blocks], e.losses.clone(), e.total.item()))
ref = runs[0]
worst = {}
for r in runs[1:]:
    for b, o, o0 in zip(e.blocks, r[0], ref[0]):
        ulp = ((o - o0).abs().max() / (o0.abs().max() * 2.0 ** -8)).item()
        worst[b["prefix"]] = max(worst.get(b["prefix"], 0.0), ulp)
for k, v in worst.items():
    print(f"{k:28s} max |diff| = {v:6.2f} bf16 ulps of the tensor max")
print("losses rel spread", max(((r[1] - ref[1]).abs() / ref[1].abs()).max().item() for r in runs[1:]), " total rel spread", max(abs(r[2] - ref[2]) / abs(ref[2]) for r in runs[1:]))
