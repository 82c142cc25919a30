#!/usr/bin/env python
"""MP-MAE pre-training driver with the reference's command line
(/root/reference/main_pretrain.py:30-162, TRAINING.md:18-42), on the MI355X HIP engine.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main_pretrain.py \
        --model convnextv2_atto --batch_size 256 --update_freq 2 --blr 1.5e-4 --epochs 200 \
        --warmup_epochs 40 --input_size 56 --patch_size 8 --norm_pix_loss True --distributed True

Dataset loading (MMEarth HDF5 / ffcv) is outside this repository's scope: batches come from
`--data_dir synthetic` (seeded synthetic tiles with the dataset's shapes, dtypes and no-data
conventions). Everything else — model construction, lr scaling, weight-decay grouping, AdamW
betas, schedule, checkpoint layout — follows the reference.
"""
import argparse
import os
import sys
import time
from pathlib import Path

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mmearth_train_amd import MODALITIES as M  # noqa: E402
from mmearth_train_amd import dist as mdist  # noqa: E402
from mmearth_train_amd import fcmae  # noqa: E402
from mmearth_train_amd.config import default_args  # noqa: E402
from mmearth_train_amd.custom_loss import UncertaintyWeightingStrategy  # noqa: E402
from mmearth_train_amd.engine_pretrain import train_one_epoch  # noqa: E402
from mmearth_train_amd.synth import make_inputs  # noqa: E402


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def get_args_parser():
    p = argparse.ArgumentParser("FCMAE pre-training", add_help=False)
    p.add_argument("--wandb", type=str2bool, default=False)
    p.add_argument("--wandb_project", type=str, default="global-lr")
    p.add_argument("--wandb_run_name", type=str)
    p.add_argument("--batch_size", default=64, type=int, help="Per GPU batch size")
    p.add_argument("--epochs", default=800, type=int)
    p.add_argument("--warmup_epochs", type=int, default=40)
    p.add_argument("--update_freq", default=1, type=int)
    p.add_argument("--loss_aggr", choices=["uncertainty", "unweighted"], default="uncertainty")
    p.add_argument("--loss_full", type=str2bool, default=False)
    p.add_argument("--model", default="convnextv2_pico", type=str)
    p.add_argument("--input_size", default=112, type=int)
    p.add_argument("--mask_ratio", default=0.6, type=float)
    p.add_argument("--norm_pix_loss", type=str2bool, default=False)
    p.add_argument("--decoder_depth", type=int, default=1)
    p.add_argument("--decoder_embed_dim", type=int, default=512)
    p.add_argument("--patch_size", type=int, default=16)
    p.add_argument("--use_orig_stem", type=str2bool, default=False)
    p.add_argument("--weight_decay", type=float, default=0.05)
    p.add_argument("--lr", type=float, default=None)
    p.add_argument("--blr", type=float, default=1.5e-4)
    p.add_argument("--min_lr", type=float, default=0.0)
    p.add_argument("--data_dir", default="synthetic", type=str)
    p.add_argument("--processed_dir", default=None, type=str)
    p.add_argument("--random_crop", type=str2bool, default=True)
    p.add_argument("--output_dir", default="")
    p.add_argument("--log_dir", default=None)
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--resume", default="")
    p.add_argument("--auto_resume", type=str2bool, default=True)
    p.add_argument("--save_ckpt", type=str2bool, default=True)
    p.add_argument("--save_ckpt_freq", default=1, type=int)
    p.add_argument("--save_ckpt_num", default=3, type=int)
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--num_workers", default=10, type=int)
    p.add_argument("--crop_pct", type=float, default=None)
    p.add_argument("--world_size", default=1, type=int)
    p.add_argument("--local-rank", default=-1, type=int)
    p.add_argument("--dist_on_itp", type=str2bool, default=False)
    p.add_argument("--dist_url", default="env://")
    p.add_argument("--use_mixed", type=str2bool, default=False)
    p.add_argument("--sparse", type=str2bool, default=True)
    p.add_argument("--debug", type=str2bool, default=False)
    p.add_argument("--distributed", type=str2bool, default=False)
    p.add_argument("--no_ffcv", type=str2bool, default=True)
    # additions of this implementation
    p.add_argument("--out_modalities", default="all_mod", help="all_mod | pix_mod | img_mod | S2 (README subsets)")
    p.add_argument("--steps_per_epoch", default=50, type=int, help="synthetic loader length")
    p.add_argument("--compute_dtype", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--fast_path", type=str2bool, default=True,
                   help="HIP-graph step runner with fused AdamW and overlapped RCCL all-reduce")
    return p


class SyntheticLoader:
    """Seeded synthetic batches with the dataset's shapes/dtypes (mmearth_dataset.py:137-142)."""

    def __init__(self, cfg, batch, steps, seed):
        self.cfg, self.batch, self.steps, self.seed = cfg, batch, steps, seed
        self.samples, _ = make_inputs(cfg, batch, seed=seed)

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            yield self.samples


from mmearth_train_amd.helpers import auto_load_model, param_groups_weight_decay, save_model  # noqa: E402,F401


def main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.distributed and world > 1:
        mdist.init(backend="nccl", local_rank=local_rank)
    device = torch.device("cuda", local_rank) if args.device == "cuda" else torch.device(args.device)
    if device.type != "cuda":
        raise RuntimeError("the pretraining path runs on the HIP engine only (no CPU fallback): --device cuda")
    if not args.sparse and args.patch_size != 16:
        raise ValueError("--sparse False: the dense encoder's stem only lines up with the patch grid at --patch_size 16 "
                         "(models/convnextv2.py:108-124; the reference fails in forward_decoder at patch 8)")
    if args.use_mixed:
        print("--use_mixed: fp16 autocast + GradScaler are replaced by bf16 activations with fp32 master weights "
              "(--compute_dtype bf16, no loss scaling needed)")
    torch.cuda.set_device(device)
    torch.manual_seed(args.seed + rank)

    a = default_args(out_modalities=M.subset(args.out_modalities), loss_aggr=args.loss_aggr,
                     use_orig_stem=args.use_orig_stem)
    for k, v in vars(a).items():
        setattr(args, k, v)
    loss_fn = UncertaintyWeightingStrategy(len(args.out_modalities)) if args.loss_aggr == "uncertainty" else None
    model = fcmae.__dict__[args.model](
        mask_ratio=args.mask_ratio, decoder_depth=args.decoder_depth, decoder_embed_dim=args.decoder_embed_dim,
        norm_pix_loss=args.norm_pix_loss, patch_size=args.patch_size, img_size=args.input_size, args=args,
        loss_fn=loss_fn, sparse=args.sparse, device=device, dtype=args.compute_dtype)
    if world > 1:   # DDP constructor semantics: parameters broadcast from rank 0
        torch.distributed.broadcast(model._pflat, src=0)
    n_parameters = sum(p.numel() for p in model.parameters() if p.requires_grad)
    if rank == 0:
        print("number of params:", n_parameters)
    eff_batch = args.batch_size * args.update_freq * world
    if args.lr is None:
        args.lr = args.blr * eff_batch / 256
    loader = SyntheticLoader(model.cfg, args.batch_size, args.steps_per_epoch, seed=1000 + rank)

    runner, optimizer = None, None
    if args.fast_path:
        # fused step: launch program + flat-buffer AdamW; --update_freq accumulates in the flat gradient buffer and
        # the bucketed RCCL all-reduce runs on the update micro-step (dist.StepRunner)
        eng = model._get_engine(args.batch_size, args.mask_ratio)
        model._engine = eng
        runner = mdist.StepRunner(eng, world_size=world, mode="program", lr=args.lr, weight_decay=args.weight_decay,
                                  update_freq=args.update_freq)
    else:
        # debugging path through torch autograd + torch.optim.AdamW; gradients are averaged across ranks on every
        # update step (engine_pretrain.train_one_epoch) - the reference wraps the model in DDP (main_pretrain.py:306-310)
        optimizer = torch.optim.AdamW(param_groups_weight_decay(model, args.weight_decay), lr=args.lr, betas=(0.9, 0.95))

    if args.output_dir:
        Path(args.output_dir).mkdir(parents=True, exist_ok=True)
    auto_load_model(args, model, optimizer=optimizer, runner=runner)       # --resume / --auto_resume (helpers.py:568-610)
    if world > 1:
        torch.distributed.broadcast(model._pflat, src=0)
    start = time.time()
    stats = None
    for epoch in range(args.start_epoch, args.epochs):
        # mask noise / crop windows are a function of (seed, rank, epoch): a resumed run repeats the uninterrupted one
        torch.manual_seed(args.seed + rank + 7919 * epoch)
        stats = train_one_epoch(model, loader, optimizer, device, epoch, args, runner=runner)
        if args.output_dir and args.save_ckpt and rank == 0 and \
                ((epoch + 1) % args.save_ckpt_freq == 0 or epoch + 1 == args.epochs):
            save_model(args, epoch, model, optimizer=optimizer, runner=runner)
    if rank == 0:
        print("Training time {:.0f}s".format(time.time() - start))
    mdist.shutdown()
    return stats


if __name__ == "__main__":
    main(get_args_parser().parse_args())
