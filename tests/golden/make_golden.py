"""Generate golden vectors from the reference's OWN modules (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--check-only]

Imports /root/reference through oracle/refharness (stubs for absent third-party packages +
a dense-backed MinkowskiEngine emulator — the ME boundary is unpinned, see
oracle/refharness/README.md), loads seeded synthetic weights (mmearth_train_amd.synth) with a
strict load_state_dict, runs forward + backward on seeded synthetic inputs, and stores small
fixtures (`tests/golden/<case>.npz`): mask noise, mask, strided slices + checksums of the
intermediate maps and predictions, all losses, and per-parameter gradient norms / slices.
Fixtures are data only; the reference's source never leaves the container.
"""
import argparse
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import mmearth_train_amd as pkg  # noqa: E402
from mmearth_train_amd import MODALITIES as M  # noqa: E402
from mmearth_train_amd.config import default_args, make_cfg  # noqa: E402
from mmearth_train_amd.synth import expand_aliases, make_inputs, make_state_dict  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

from tests.golden_cases import (CASES, GRAD_SLICES, case_cfg, case_data, checks,  # noqa: E402
                                strided)


def run_reference(c, cfg, sd, inputs, noise_seed):
    from oracle.refharness.load_reference import load
    ref = load()
    args = default_args(out_modalities=M.subset(c["subset"]), loss_aggr=c["aggr"], use_orig_stem=c.get("orig_stem", False))
    loss_fn = (ref.custom_loss.UncertaintyWeightingStrategy(len(cfg.out_mods))
               if c["aggr"] == "uncertainty" else None)
    model = ref.fcmae.__dict__[c["model"]](
        mask_ratio=0.6, decoder_depth=c.get("decoder_depth", 1), decoder_embed_dim=512, norm_pix_loss=c["norm_pix"],
        patch_size=c["patch"], img_size=c["img"], args=args, loss_fn=loss_fn, sparse=c.get("sparse", True))
    full = expand_aliases(cfg, sd)
    missing = model.load_state_dict(full, strict=True)
    taps = {}
    def _tap(name):
        def hook(mod, inp, outp):
            taps[name] = outp
            if name == "dec_out":
                taps["dec_in"] = inp[0]
        return hook

    hooks = [
        model.encoder.register_forward_hook(_tap("enc_out")),
        model.decoder_dict[cfg.out_mods[0].name].register_forward_hook(_tap("dec_out")),
    ]
    torch.manual_seed(noise_seed)   # first RNG draw in forward = randn(N, L) of gen_random_mask
    out = model({k: v.clone() for k, v in inputs.items()}, mask_ratio=0.6)
    out[0].backward()
    for h in hooks:
        h.remove()
    named = dict(model.named_parameters())   # shared tensors appear under their first alias
    grads = OrderedDict()
    for k in sd:
        p = named[k]
        grads[k] = p.grad if p.grad is not None else torch.zeros_like(p)
    return out, taps, grads


def build_fixture(c, cfg, noise, out, taps, grads):
    loss, pred, mask, loss_dict, log_vars, weighted = out
    fx = OrderedDict()
    fx["noise"] = noise.numpy()
    fx["mask"] = mask.detach().numpy()
    fx["loss"] = np.array(loss.item(), dtype=np.float64)
    fx["loss_dict"] = np.array([v.item() for v in loss_dict.values()], dtype=np.float64)
    if weighted is not None:
        fx["weighted"] = weighted.detach().double().numpy()
        fx["log_vars"] = np.array(log_vars, dtype=np.float64)
    fx["enc_out_s"] = strided(taps["enc_out"], 3)
    fx["enc_out_c"] = checks(taps["enc_out"])
    for k in ("dec_in", "dec_out"):
        fx[k + "_s"] = strided(taps[k], 7)
        fx[k + "_c"] = checks(taps[k])
    for om in cfg.out_mods:
        p = pred[om.name]
        fx[f"pred_{om.name}_s"] = strided(p, 23 if p.numel() > 4096 else 1)
        fx[f"pred_{om.name}_c"] = checks(p)
    keys = list(grads.keys())
    fx["grad_norms"] = np.array([grads[k].double().norm().item() for k in keys], dtype=np.float64)
    fx["grad_sums"] = np.array([grads[k].double().sum().item() for k in keys], dtype=np.float64)
    for k, sl in GRAD_SLICES.items():
        if k in grads:
            g = grads[k]
            g2 = g.reshape(g.shape[0], -1) if g.dim() > 2 and len(sl) == 2 else g
            fx["grad:" + k] = g2[sl].detach().float().numpy().copy()
    return fx


def compare_with_oracle(c, cfg, sd, inputs, noise, out, taps, grads):
    from oracle import mpmae_ref as O
    otaps = {}
    p = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in sd.items())
    oout = O.forward(p, inputs, noise, cfg, taps=otaps)
    oout[0].backward()
    worst = 0.0

    def rel(a, b):
        a, b = a.detach().double(), b.detach().double()
        return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()

    assert torch.equal(oout[2], out[2]), "mask mismatch"
    rows = [("loss", rel(oout[0], out[0]))]
    for k in ("enc_out", "dec_in", "dec_out"):
        rows.append((k, rel(otaps[k], taps[k])))
    for om in cfg.out_mods:
        rows.append(("pred_" + om.name, rel(oout[1][om.name], out[1][om.name])))
        rows.append(("loss_" + om.name, rel(oout[3][om.name], out[3][om.name])))
    gworst, gkey = 0.0, None
    for k in sd:
        g = p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])
        r = rel(g, grads[k]) if grads[k].abs().max() > 0 else g.abs().max().item()
        if r > gworst:
            gworst, gkey = r, k
    rows.append((f"grad(worst: {gkey})", gworst))
    for name, r in rows:
        worst = max(worst, r)
    print(f"  oracle vs reference [{c_name(c)}]: worst rel err {worst:.3e}; "
          + ", ".join(f"{n}={r:.1e}" for n, r in rows if r == worst or n in ("loss", "enc_out")))
    return worst


def c_name(c):
    for k, v in CASES.items():
        if v is c:
            return k
    return "?"


def misc_fixture():
    """Small known-answer values for the schedule and checkpoint remap (helpers.py:647-707)."""
    from oracle.refharness.load_reference import load
    ref = load()
    from argparse import Namespace
    fx = OrderedDict()
    a = Namespace(lr=2.4e-3, min_lr=0.0, warmup_epochs=40, epochs=200)

    class _Opt:
        param_groups = [{"lr": 0.0}]
    es = [0.0, 0.5, 39.99, 40.0, 77.3, 120.0, 199.5]
    fx["lr_epochs"] = np.array(es)
    fx["lr_values"] = np.array([ref.helpers.adjust_learning_rate(_Opt(), e, a) for e in es])
    cfg = make_cfg()
    sd = expand_aliases(cfg, make_state_dict(cfg, seed=3))
    enc = OrderedDict((k, v) for k, v in sd.items() if k.startswith("encoder."))
    rm = ref.helpers.remap_checkpoint_keys(enc)
    fx["remap_keys"] = np.array(list(rm.keys()))
    fx["remap_shapes"] = np.array([";".join(map(str, v.shape)) for v in rm.values()])
    fx["remap_sums"] = np.array([v.double().sum().item() for v in rm.values()])
    k = "downsample_layers.0.1.weight"
    fx["remap_sample"] = rm[k][::8, ::8].contiguous().numpy()
    return fx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--cases", nargs="*", default=list(CASES))
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    worst = 0.0
    for name in a.cases:
        c = CASES[name]
        cfg = case_cfg(c)
        sd, inputs, noise = case_data(c, cfg)
        out, taps, grads = run_reference(c, cfg, sd, inputs, c["nseed"])
        print(f"{name}: reference loss {out[0].item():.6f}")
        worst = max(worst, compare_with_oracle(c, cfg, sd, inputs, noise, out, taps, grads))
        if not a.check_only:
            fx = build_fixture(c, cfg, noise, out, taps, grads)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    if not a.check_only:
        np.savez_compressed(os.path.join(HERE, "misc.npz"), **misc_fixture())
    print(f"worst oracle-vs-reference relative error over all cases: {worst:.3e}")
    for p, _, _ in os.walk("/root/reference"):
        assert "__pycache__" not in p, "bytecode leaked into /root/reference"


if __name__ == "__main__":
    main()
