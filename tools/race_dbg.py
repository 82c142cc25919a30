import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.chdir("/root/repo")
from test_hip_parity import CASES, case_cfg, case_data, _engine, _rel
from mmearth_train_amd import dist as mdist
c = CASES["allmod_atto_56"]; cfg = case_cfg(c)
sd, inputs, noise = case_data(c, cfg)
out = {}
for trial in range(6):
    for m in ("eager", "program"):
        eng = _engine(cfg, c["N"], "f32", sd, inputs, noise, block_mode="mat", lanes=(m != "eager"))
        run = mdist.StepRunner(eng, world_size=1, lr=1e-3, mode=m)
        for st in range(int(os.environ.get("NSTEPS", "3"))):
            run.step()
        torch.cuda.synchronize()
        out[m] = ({k: eng.grads[k].clone() for k in eng.grads}, eng.losses.clone(), {k: eng.params[k].clone() for k in eng.params})
    ge, gp = out["eager"][0], out["program"][0]
    bad = [(k, round(_rel(ge[k], gp[k]), 5)) for k in ge if _rel(ge[k], gp[k]) > 1e-4]
    pe, pp = out["eager"][2], out["program"][2]
    badp = [(k, round(_rel(pe[k], pp[k]), 6)) for k in pe if _rel(pe[k], pp[k]) > 1e-5]
    print("trial", trial, "loss rel", _rel(out["eager"][1], out["program"][1]), "bad grads:", len(bad), bad[:6], "bad params:", len(badp), badp[:6], flush=True)
