import sys, os, torch
sys.path.insert(0, "/root/repo")
from tests.golden_cases import CASES, case_cfg, case_data
from tests.test_hip_parity import _engine, _rel
c = CASES["allmod_atto_56"]; cfg = case_cfg(c)
sd, inputs, noise = case_data(c, cfg)
junk = [torch.randn(1 << 24, device="cuda") * 100 for _ in range(8)]; del junk     # poison the allocator's free blocks
for it in range(6):
    for dtype in ("bf16", "fp8"):
        eng = _engine(cfg, c["N"], dtype, sd, inputs, noise)
        eng.run_segment("encoder"); torch.cuda.synchronize()
        with torch.no_grad():
            eng.params["proj.weight"].mul_(1.5); eng.params["decoder_dict.sentinel2.0.pwconv1.weight"].mul_(0.5)
        eng.run_segment("decoder"); torch.cuda.synchronize()
        got = {k: v.float().clone() for k, v in eng.preds().items()}
        enc1 = eng.enc_out.float().clone() if hasattr(eng, "enc_out") else None
        eng.forward(); torch.cuda.synchronize()
        full = {k: v.float().clone() for k, v in eng.preds().items()}
        eng.forward(); torch.cuda.synchronize()
        full2 = {k: v.float().clone() for k, v in eng.preds().items()}
        eng.run_segment("decoder"); torch.cuda.synchronize()
        got2 = {k: v.float().clone() for k, v in eng.preds().items()}
        w = max((_rel(got[k], full[k]), k) for k in full)
        w2 = max((_rel(full2[k], full[k]), k) for k in full)
        w3 = max((_rel(got2[k], full[k]), k) for k in full)
        print(it, dtype, "seg-vs-full", w, "| full-vs-full", w2, "| seg-after-full vs full", w3, flush=True)
    junk = [torch.randn(1 << 22, device="cuda") * 1000 for _ in range(it + 1)]; del junk
