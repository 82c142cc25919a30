"""How many cross-lane events one step records / waits for (launch-program structure of Engine at the bench config)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmearth_train_amd.config import make_cfg
from mmearth_train_amd.engine import Engine
eng = Engine(make_cfg(), 256, dtype="bf16", device="cuda:0")
for nm, ops in (("fwd", eng.fwd_ops), ("bwd", eng.bwd_ops)):
    main = [o for o in ops if o[3]["lane"] == 0]
    side = [o for o in ops if o[3]["lane"] != 0]
    print(nm, "main ops", len(main), "signals", sum(o[3]["signal"] is not None for o in main), "waits", sum(len(o[3]["wait"]) for o in main),
          "| side ops", len(side), "signals", sum(o[3]["signal"] is not None for o in side), "waits", sum(len(o[3]["wait"]) for o in side))
for o in eng.bwd_ops[:60]:
    m = o[3]
    print(f"  lane {m['lane']} {o[0]:55s} wait={list(m['wait'])} signal={m['signal']}")
