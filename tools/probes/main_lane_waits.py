"""Which main-lane ops of the backward wait for a side-lane signal (and which side-lane op raises it)? Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmearth_train_amd.config import make_cfg
from mmearth_train_amd.engine import Engine
e = Engine(make_cfg(), 256, dtype="bf16", device="cuda:0")
for nm, ops in (("fwd", e.fwd_ops), ("bwd", e.bwd_ops)):
    sig = {}
    for i, (name, fn, args, m) in enumerate(ops):
        if m["signal"]:
            sig[m["signal"]] = (i, name, m["lane"])
    for i, (name, fn, args, m) in enumerate(ops):
        for w in m["wait"]:
            if w in sig and sig[w][2] != m["lane"] and m["lane"] == 0:
                print(f"{nm} op {i:3d} {name:50s} (lane 0) waits for {w} <- op {sig[w][0]:3d} {sig[w][1]} (lane {sig[w][2]})")
            elif w not in sig and m["lane"] == 0:
                print(f"{nm} op {i:3d} {name:50s} (lane 0) waits for {w} (raised in another list)")
