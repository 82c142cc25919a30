"""Checkpoint contract of the pretraining path (/root/reference/helpers.py:529-610 save_model /
auto_load_model, :668-707 remap_checkpoint_keys, :404-466 load_state_dict) for the HIP engine.

A checkpoint is `{"model", "optimizer", "epoch", "scaler", "args"}` (`helpers.py:543-549`). With the fused
AdamW the moments live in two flat fp32 buffers; they are written in **torch.optim.AdamW's own state_dict
layout** (two timm weight-decay groups, per-parameter `step / exp_avg / exp_avg_sq`), so the file is loadable
both by this repository's runner and by a stock `torch.optim.AdamW` built the reference's way
(`main_pretrain.py:312-320`) - which is what `helpers.auto_load_model` does with the "optimizer" entry.
"""
import glob
import math
import os
from collections import OrderedDict
from pathlib import Path

import torch


def _decay_split(model):
    """timm.optim.optim_factory.param_groups_weight_decay order: [no_decay names], [decay names]."""
    no_decay, decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.ndim <= 1 or name.endswith(".bias")) else decay).append(name)
    return no_decay, decay


def param_groups_weight_decay(model, weight_decay):
    no_decay, decay = _decay_split(model)
    named = dict(model.named_parameters())
    return [{"params": [named[n] for n in no_decay], "weight_decay": 0.0},
            {"params": [named[n] for n in decay], "weight_decay": weight_decay}]


def fused_adamw_state_dict(model, runner):
    """The runner's fused AdamW state in torch.optim.AdamW.state_dict() layout."""
    eng = runner.eng
    no_decay, decay = _decay_split(model)
    order = no_decay + decay
    state = {}
    if runner.t > 0:
        m, v = eng.mflat.detach().cpu(), eng.vflat.detach().cpu()
        for i, name in enumerate(order):
            off, n = eng.offsets[name]
            shape = tuple(eng.params[name].shape)
            state[i] = {"step": torch.tensor(float(runner.t)), "exp_avg": m[off:off + n].view(shape).clone(),
                        "exp_avg_sq": v[off:off + n].view(shape).clone()}
    base = dict(lr=runner.lr, betas=(0.9, 0.95), eps=1e-8, amsgrad=False, maximize=False, foreach=None, capturable=False,
                differentiable=False, fused=None, decoupled_weight_decay=True)
    groups = [dict(base, weight_decay=0.0, params=list(range(len(no_decay)))),
              dict(base, weight_decay=runner.wd, params=list(range(len(no_decay), len(order))))]
    return {"state": state, "param_groups": groups}


def load_fused_adamw_state_dict(model, runner, sd):
    """Inverse of fused_adamw_state_dict (also accepts what torch.optim.AdamW.state_dict() wrote for the same groups)."""
    eng = runner.eng
    no_decay, decay = _decay_split(model)
    order = no_decay + decay
    groups = sd["param_groups"]
    if sum(len(g["params"]) for g in groups) != len(order):
        raise ValueError("optimizer state does not match this model's parameter groups")
    step = 0
    eng.mflat.zero_()
    eng.vflat.zero_()
    for i, name in enumerate(order):
        st = sd["state"].get(i)
        if st is None:
            continue
        off, n = eng.offsets[name]
        eng.mflat[off:off + n].copy_(st["exp_avg"].reshape(-1))
        eng.vflat[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
        step = max(step, int(float(st["step"])))
    runner.t, runner.micro = step, 0


def save_model(args, epoch, model, optimizer=None, runner=None):
    """helpers.save_model (:529-565): rank-0 write of checkpoint-<epoch>.pth, rolling deletion."""
    out = Path(args.output_dir)
    opt_state = fused_adamw_state_dict(model, runner) if runner is not None else optimizer.state_dict()
    torch.save({"model": OrderedDict((k, v.detach().cpu().clone()) for k, v in model.state_dict().items()),
                "optimizer": opt_state, "epoch": epoch, "scaler": {}, "args": args},
               out / f"checkpoint-{epoch}.pth")
    if isinstance(epoch, int):
        old = out / f"checkpoint-{epoch - args.save_ckpt_num * args.save_ckpt_freq}.pth"
        if old.exists():
            os.remove(old)


def auto_load_model(args, model, optimizer=None, runner=None):
    """helpers.auto_load_model (:568-610): newest checkpoint-*.pth of output_dir when --auto_resume and no
    --resume; loads model (strict), optimizer state, sets args.start_epoch. Returns the path or None."""
    if getattr(args, "auto_resume", False) and len(args.resume or "") == 0 and args.output_dir:
        latest = -1
        for ck in glob.glob(os.path.join(args.output_dir, "checkpoint-*.pth")):
            t = ck.split("-")[-1].split(".")[0]
            if t.isdigit():
                latest = max(int(t), latest)
        if latest >= 0:
            args.resume = os.path.join(args.output_dir, "checkpoint-%d.pth" % latest)
        print("Auto resume checkpoint: %s" % args.resume)
    if not args.resume:
        return None
    ck = torch.load(args.resume, map_location="cpu", weights_only=False)
    model.load_state_dict(ck["model"])
    print("Resume checkpoint %s" % args.resume)
    if "optimizer" in ck and "epoch" in ck and ck["optimizer"] is not None:
        if runner is not None:
            load_fused_adamw_state_dict(model, runner, ck["optimizer"])
        elif optimizer is not None:
            optimizer.load_state_dict(ck["optimizer"])
        if not isinstance(ck["epoch"], str):
            args.start_epoch = ck["epoch"] + 1
        print("With optim & sched!")
    return args.resume


def remap_checkpoint_keys(ckpt):
    """Sparse-encoder checkpoint -> dense ConvNeXtV2 keys/layouts (helpers.py:668-707): drop the `encoder.`
    prefix; ME kernels (k*k, Cin, Cout) -> conv weights [Cout, Cin, kh, kw] with ME's kernel index running the
    FIRST spatial coordinate fastest (W[o, i, kh, kw] = K[kw*ks + kh, i, o]); depthwise (k*k, C) -> [C, 1, kh, kw];
    `.ln.` / `.linear.` wrappers removed; (1, C) biases flattened; GRN (1, 4C) -> (1, 1, 1, 4C)."""
    out = OrderedDict()
    for k, v in ckpt.items():
        if k.startswith("encoder"):
            k = k.split(".", 1)[1]
        if k.endswith("kernel"):
            base = k.rsplit(".", 1)[0] + ".weight"
            ks = int(math.sqrt(v.shape[0]))
            if v.dim() == 3:
                out[base] = v.permute(2, 1, 0).reshape(v.shape[2], v.shape[1], ks, ks).transpose(3, 2)
            elif v.dim() == 2:
                out[base] = v.permute(1, 0).reshape(v.shape[1], 1, ks, ks).transpose(3, 2)
            continue
        if "ln" in k or "linear" in k:
            parts = k.split(".")
            parts.pop(-2)
            k = ".".join(parts)
        elif "backbone.resnet" in k:
            k = k.split("backbone.resnet.")[1]
        out[k] = v
    for k, v in out.items():
        if k.endswith("bias") and v.dim() != 1:
            out[k] = v.reshape(-1)
        elif "grn" in k:
            out[k] = v.unsqueeze(0).unsqueeze(1)
    return out


def load_state_dict(model, state_dict, prefix="", ignore_missing="relative_position_index"):
    """Non-strict load with the reference's reporting (helpers.py:404-466)."""
    res = model.load_state_dict(state_dict, strict=False)
    missing = [k for k in res.missing_keys if not any(ig in k for ig in ignore_missing.split("|"))]
    if missing:
        print("Weights of {} not initialized from pretrained model: {}".format(model.__class__.__name__, missing))
    if res.unexpected_keys:
        print("Weights from pretrained model not used in {}: {}".format(model.__class__.__name__, res.unexpected_keys))
    return res
