"""Pretraining epoch loop (drop-in for /root/reference/engine_pretrain.py:21-122) without the
reference's 14 host synchronisations per iteration: losses stay on the device and are read back
only every `print_freq` iterations."""
import math
import sys
import time

import torch

from . import dist as mdist


def adjust_learning_rate(optimizer, epoch, args):
    """Warm-up then half-cycle cosine, per iteration (/root/reference/helpers.py:647-665)."""
    if epoch < args.warmup_epochs:
        lr = args.lr * epoch / args.warmup_epochs
    else:
        lr = args.min_lr + (args.lr - args.min_lr) * 0.5 * (
            1.0 + math.cos(math.pi * (epoch - args.warmup_epochs) / (args.epochs - args.warmup_epochs)))
    if optimizer is not None:
        for g in optimizer.param_groups:
            g["lr"] = lr * g["lr_scale"] if "lr_scale" in g else lr
    return lr


def _allreduce_grads_(model):
    """DDP gradient averaging for the torch-autograd path (main_pretrain.py:306-310)."""
    import torch.distributed as tdist
    if not tdist.is_initialized() or tdist.get_world_size() == 1:
        return
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    tdist.all_reduce(flat)
    flat /= tdist.get_world_size()
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def train_one_epoch(model, data_loader, optimizer, device, epoch, args, runner=None, print_freq=20):
    """model: fcmae.FCMAE. data_loader yields dicts modality -> tensor. With `runner` (dist.StepRunner) the
    fused launch program (forward, backward, bucketed all-reduce, AdamW; gradient accumulation over
    args.update_freq micro-steps) is used; otherwise torch autograd + `optimizer` with explicit gradient
    averaging across ranks. A non-finite loss stops training (engine_pretrain.py:83-85): on the fused path the
    device skips the poisoned update itself (mpmae_hp_fetch) and the host notices at the next read-back."""
    update_freq = args.update_freq
    n_iter = len(data_loader)
    t0 = time.time()
    loss_value, loss_dict, log_vars, normalized = float("nan"), {}, None, None
    if optimizer is not None:
        optimizer.zero_grad()
    if runner is not None:
        runner.eng.reset_meters()          # device-resident SmoothedValue(window 20) meters, written by the optimizer launch itself
    for it, samples in enumerate(data_loader):
        lr = adjust_learning_rate(optimizer, it / n_iter + epoch, args) if it % update_freq == 0 else None
        if runner is not None:
            if lr is not None:
                runner.lr = lr
            eng = runner.eng
            # input stage of this batch on its own stream, overlapped with the previous step's backward: host-to-device copies, the
            # crop windows, crop / copy into the engine's buffers and the mask noise are all issued inside the stage
            with eng.input_stage(runner):
                samples = {k: v.to(device, non_blocking=True) for k, v in samples.items()}
                eng.set_inputs(samples, None, crop=model._crop_windows(samples))
            runner.step()
            losses_t, total_t = eng.losses, eng.total
        else:
            samples = {k: v.to(device, non_blocking=True) for k, v in samples.items()}
            loss, pred, mask, loss_dict_, log_vars, normalized = model(samples, mask_ratio=args.mask_ratio)
            if not math.isfinite(float(loss.item())):
                print("Loss is {}, stopping training".format(float(loss.item())))
                sys.exit(1)
            (loss / update_freq).backward()
            if (it + 1) % update_freq == 0:
                _allreduce_grads_(model)
                optimizer.step()
                optimizer.zero_grad()
            losses_t, total_t = model._engine.losses, loss.detach()
        if runner is not None and (it % print_freq == 0 or it == n_iter - 1):
            # the only host sync of the fused loop: ONE copy of the meter ring (loss, 12 per-modality losses, their weighted forms, the
            # gradient norm; median / window average / global average as in MetricLogger.log_every, helpers.py:147-206)
            m = runner.eng.read_meters()
            if "loss" in m:
                loss_value = m["loss"]["value"]
                if runner.barrier_timeouts() > 0:
                    raise RuntimeError("a persistent stage kernel timed out at its grid barrier (a workgroup never became resident - is another "
                                       "process using this GPU?); the update was skipped. Re-run with Engine option ps=0 to use the per-block kernels")
                if not math.isfinite(loss_value) or runner.skipped_steps() > 0:
                    print("Loss is {}, stopping training".format(loss_value))
                    sys.exit(1)
                names = [om.name for om in model.cfg.out_mods]
                loss_dict = {n: m[f"loss_{n}"]["value"] for n in names}
                if model.cfg.loss_aggr == "uncertainty":
                    log_vars = model.loss_fn.log_vars.tolist()
                    normalized = torch.tensor([m[f"weighted_{n}"]["value"] for n in names])
                gn = m.get("grad_norm")
                print(f"Epoch: [{epoch}]  [{it}/{n_iter}]  loss: {m['loss']['median']:.4f} ({m['loss']['global_avg']:.4f})  "
                      + (f"grad_norm: {gn['median']:.4f} ({gn['global_avg']:.4f})  " if gn else "")
                      + f"img/s: {(it + 1) * samples['sentinel2'].shape[0] / (time.time() - t0):.0f}", flush=True)
        elif runner is None and (it % print_freq == 0 or it == n_iter - 1):
            loss_value = float(total_t.item())
            if not math.isfinite(loss_value) or (runner is not None and runner.skipped_steps() > 0):
                print("Loss is {}, stopping training".format(loss_value))
                sys.exit(1)
            names = [om.name for om in model.cfg.out_mods]
            loss_dict = dict(zip(names, losses_t.tolist()))
            if model.cfg.loss_aggr == "uncertainty":
                log_vars = model.loss_fn.log_vars.tolist()
                normalized = model._engine.weighted.clone() if model._engine is not None else None
            mean = mdist.mean_scalar(loss_value)
            print(f"Epoch: [{epoch}]  [{it}/{n_iter}]  loss: {mean:.4f}  "
                  f"img/s: {(it + 1) * samples['sentinel2'].shape[0] / (time.time() - t0):.0f}", flush=True)
    if runner is not None:
        # "Averaged stats" of the epoch, one fused all-reduce across the ranks (engine_pretrain.py:115-118)
        g = runner.eng.meter_global_averages()
        names = [om.name for om in model.cfg.out_mods]
        loss_dict = {n: g[f"loss_{n}"] for n in names}
        if model.cfg.loss_aggr == "uncertainty":
            normalized = torch.tensor([g[f"weighted_{n}"] for n in names])
        return {"loss": g["loss"], "grad_norm": g["grad_norm"], "loss_last": loss_value}, loss_dict, log_vars, normalized
    return {"loss": loss_value}, loss_dict, log_vars, normalized
