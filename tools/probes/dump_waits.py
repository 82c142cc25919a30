import sys
sys.path.insert(0,".")
from mmearth_train_amd.config import make_cfg
from mmearth_train_amd.engine import Engine
cfg=make_cfg()
e=Engine(cfg,256,dtype="bf16",device="cuda")
ops=e.bwd_ops
names=[o[0] for o in ops]
sig={o[3]["signal"]:o[0] for o in ops if o[3]["signal"]}
for key in ("encoder.downsample_layers.2:dgrad","encoder.downsample_layers.1:dgrad","encoder.downsample_layers.0:dgrad"):
    i=names.index(key)
    for j in range(i-6,i+3):
        o=ops[j]; print(j, o[3]["lane"], o[0], "wait",[ (w, sig.get(w)) for w in o[3]["wait"]], "sig",o[3]["signal"])
    print()
