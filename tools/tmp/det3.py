import sys, os, torch
sys.path.insert(0, "/root/repo")
from tests.golden_cases import CASES, case_cfg, case_data
from tests.test_hip_parity import _engine, _rel
c = CASES["allmod_atto_56"]; cfg = case_cfg(c)
sd, inputs, noise = case_data(c, cfg)
opts = dict(kv.split("=") for kv in sys.argv[1].split(",")) if len(sys.argv) > 1 and sys.argv[1] else {}
opts = {k: int(v) for k, v in opts.items()}
from mmearth_train_amd.engine import Engine
for dtype in ("bf16", "fp8"):
    eng = Engine(cfg, c["N"], dtype=dtype, device="cuda:0", options=opts)
    eng.load_state_dict(sd); eng.set_inputs(inputs, noise)
    names = ["x0", "yln", "pooled", "pred_img", "pred_pix", "yhat", "rstd_y"]
    ref = None; nbad = 0
    for it in range(300):
        eng.forward(); torch.cuda.synchronize()
        cur = {k: getattr(eng, k).float().clone() for k in names}
        cur["y"] = eng.dec["out"].float().clone() if hasattr(eng, "dec") and "out" in eng.dec else cur["yln"]
        if ref is None: ref = cur; continue
        bad = {k: _rel(cur[k], ref[k]) for k in cur if not torch.equal(cur[k], ref[k])}
        if bad:
            nbad += 1
            if nbad <= 5: print(dtype, "iter", it, bad, flush=True)
    print(dtype, opts, "differing forwards:", nbad, "of 299", flush=True)
