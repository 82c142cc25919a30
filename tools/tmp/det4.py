import sys, os, torch
sys.path.insert(0, "/root/repo")
from tests.golden_cases import CASES, case_cfg, case_data
from tests.test_hip_parity import _rel
c = CASES["allmod_atto_56"]; cfg = case_cfg(c)
sd, inputs, noise = case_data(c, cfg)
opts = dict(kv.split("=") for kv in sys.argv[1].split(",")) if len(sys.argv) > 1 and sys.argv[1] else {}
opts = {k: int(v) for k, v in opts.items()}
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp8"
from mmearth_train_amd.engine import Engine
eng = Engine(cfg, c["N"], dtype=dtype, device="cuda:0", options=opts)
eng.load_state_dict(sd); eng.set_inputs(inputs, noise)
def snap():
    d = {"x0": eng.x0, "act0": eng.act[0], "act1": eng.act[1], "act2": eng.act[2], "act3": eng.act[3], "vis": eng.vis, "inv": eng.inv}
    for i, b in enumerate(eng.blocks):
        for k in ("dd", "dw", "xn", "dhat", "rstd", "h", "G2", "Gx", "scale", "z", "out"):
            if k in b and isinstance(b[k], torch.Tensor): d[f"b{i}.{k}"] = b[k]
    for i, dn in enumerate(eng.down):
        for k, v in dn.items():
            if isinstance(v, torch.Tensor): d[f"down{i}.{k}"] = v
    return {k: v.clone() for k, v in d.items()}
print("ops:", [op[0] + ("@1" if op[3]["lane"] else "") for op in eng.fwd_ops][:24])
ref = None; nbad = 0
for it in range(200):
    eng.forward(); torch.cuda.synchronize()
    cur = snap()
    if ref is None: ref = cur; print(sorted(cur.keys())[:60]); continue
    bad = [k for k in cur if not torch.equal(cur[k], ref[k])]
    if bad:
        nbad += 1
        if nbad <= 4: print("iter", it, "first differing:", bad[:12], flush=True)
print(dtype, opts, "differing forwards:", nbad)
