"""Developer tool: engine (GPU) vs oracle (CPU) on a golden case, stage by stage."""
import argparse
import sys
import os
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mpmae_ref as O  # noqa: E402
from tests.golden_cases import CASES, case_cfg, case_data  # noqa: E402
from mmearth_train_amd.engine import Engine  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="allmod_atto_56")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--no-bwd", action="store_true")
    ap.add_argument("--block-mode", default=None)
    a = ap.parse_args()
    c = CASES[a.case]
    cfg = case_cfg(c)
    sd, inputs, noise = case_data(c, cfg)
    taps = {}
    p = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    loss, pred, mask, loss_dict, log_vars, weighted = O.forward(p, inputs, noise, cfg, taps=taps)
    loss.backward()

    eng = Engine(cfg, c["N"], dtype=a.dtype, block_mode=a.block_mode)
    eng.load_state_dict(sd)
    eng.set_inputs(inputs, noise)
    eng.forward()
    torch.cuda.synchronize()
    print("mask equal:", torch.equal(eng.mask.cpu(), mask))
    C = cfg.dims
    print("stem_out", rel(eng.dense_map(eng.x0, C[0], 0), taps["stem_out"]))
    bi = 0
    for i in range(4):
        bi += cfg.depths[i]
        print(f"stage{i}_out", rel(eng.dense_map(eng.blocks[bi - 1]["out"], C[i], i), taps[f"stage{i}_out"]))
    N, L, D, g = eng.N, eng.L, eng.D, eng.grid
    xd = eng.xdec.float().reshape(N, g, g, D).permute(0, 3, 1, 2)
    print("dec_in", rel(xd, taps["dec_in"]))
    yd = eng.dec_out.float().reshape(N, g, g, D).permute(0, 3, 1, 2)
    print("dec_out", rel(yd, taps["dec_out"]))
    pr = eng.preds()
    for om in cfg.out_mods:
        print("pred", om.name, rel(pr[om.name].float(), pred[om.name]))
    print("losses eng", [round(v, 5) for v in eng.losses.tolist()])
    print("losses ora", [round(v.item(), 5) for v in loss_dict.values()])
    print("total", eng.total.item(), loss.item())
    if a.no_bwd:
        return
    eng.backward()
    torch.cuda.synchronize()
    worst = []
    for k in sd:
        g_o = p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])
        g_e = eng.grads[k].cpu()
        den = g_o.abs().max().item()
        r = (g_e - g_o).abs().max().item() / (den + 1e-30) if den > 0 else g_e.abs().max().item()
        worst.append((r, k))
    worst.sort(reverse=True)
    print("grad worst 25:")
    for r, k in worst[:25]:
        print(f"  {r:.3e}  {k}")
    print("grad median rel err:", sorted(r for r, _ in worst)[len(worst) // 2])


if __name__ == "__main__":
    main()
