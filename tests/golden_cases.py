"""Golden-case definitions shared by tests/golden/make_golden.py (which produced the
fixtures from the reference) and the parity tests (which regenerate the same seeded weights
and inputs and compare against the stored reference outputs)."""
import os
from collections import OrderedDict

import numpy as np
import torch

from mmearth_train_amd import MODALITIES as M
from mmearth_train_amd.config import make_cfg
from mmearth_train_amd.synth import make_inputs, make_state_dict

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> dict(model, img, patch, subset, N, norm_pix, aggr, wseed, iseed, nseed, zero_pix)
CASES = OrderedDict(
    allmod_atto_56=dict(model="convnextv2_atto", img=56, patch=8, subset="all_mod", N=2,
                        norm_pix=True, aggr="uncertainty", wseed=11, iseed=21, nseed=31),
    s2_atto_56_bs4=dict(model="convnextv2_atto", img=56, patch=8, subset="S2", N=4,
                        norm_pix=True, aggr="uncertainty", wseed=12, iseed=22, nseed=32),
    allmod_atto_56_unweighted=dict(model="convnextv2_atto", img=56, patch=8, subset="all_mod", N=2,
                                   norm_pix=False, aggr="unweighted", wseed=13, iseed=23, nseed=33),
    pixmod_atto_56=dict(model="convnextv2_atto", img=56, patch=8, subset="pix_mod", N=2,
                        norm_pix=True, aggr="uncertainty", wseed=14, iseed=24, nseed=34),
    allmod_tiny_112=dict(model="convnextv2_tiny", img=112, patch=16, subset="all_mod", N=2,
                         norm_pix=True, aggr="uncertainty", wseed=15, iseed=25, nseed=35),
    allmod_atto_56_zeropix=dict(model="convnextv2_atto", img=56, patch=8, subset="all_mod", N=2,
                                norm_pix=True, aggr="uncertainty", wseed=16, iseed=26, nseed=36,
                                zero_pix=True),
    # decoder_depth = 2 (fcmae.py:119-121: an nn.Sequential of decoder_depth Blocks shared by every modality)
    allmod_atto_56_dec2=dict(model="convnextv2_atto", img=56, patch=8, subset="all_mod", N=2,
                             norm_pix=True, aggr="uncertainty", wseed=17, iseed=27, nseed=37, decoder_depth=2),
    # FCMAE(sparse=False): the dense ConvNeXtV2 encoder (fcmae.py:103-111), the mode the reference's own test runs
    # (tests/pretrain_test.py:17); its stem only lines up at patch 16 (convnextv2.py:108-124)
    allmod_atto_112_dense=dict(model="convnextv2_atto", img=112, patch=16, subset="all_mod", N=2,
                               norm_pix=True, aggr="uncertainty", wseed=18, iseed=28, nseed=38, sparse=False),
    # use_orig_stem=True (convnextv2_sparse.py:99-110,202-203 / convnextv2.py:97-106): one convolution k = s = patch / 8 + LN instead of
    # initial_conv + depthwise stem; k = 1 at patch 8 (ME kernel (Cin, C0)), k = 2 at patch 16, and the dense encoder's Conv2d form
    allmod_atto_56_origstem=dict(model="convnextv2_atto", img=56, patch=8, subset="all_mod", N=2,
                                 norm_pix=True, aggr="uncertainty", wseed=19, iseed=29, nseed=39, orig_stem=True, zero_pix=True),
    allmod_atto_112_origstem=dict(model="convnextv2_atto", img=112, patch=16, subset="all_mod", N=2,
                                  norm_pix=True, aggr="uncertainty", wseed=20, iseed=30, nseed=40, orig_stem=True, zero_pix=True),
    allmod_atto_112_dense_origstem=dict(model="convnextv2_atto", img=112, patch=16, subset="all_mod", N=2,
                                        norm_pix=True, aggr="uncertainty", wseed=21, iseed=31, nseed=41, sparse=False, orig_stem=True),
)

GRAD_SLICES = {
    "encoder.initial_conv.0.kernel": (slice(None), slice(None), slice(None, None, 4)),
    "encoder.stem_orig.0.kernel": (Ellipsis, slice(None, None, 4)),
    "encoder.stem_orig.0.weight": (slice(None, None, 4), slice(None)),
    "encoder.stages.2.3.grn.gamma": (slice(None), slice(None, None, 4)),
    "encoder.stages.0.1.dwconv.kernel": (slice(None), slice(None, None, 4)),
    "encoder.downsample_layers.1.1.kernel": (slice(None), slice(None, None, 8), slice(None, None, 8)),
    "encoder.initial_conv.0.weight": (slice(None, None, 4), slice(None)),
    "encoder.stem.0.weight": (slice(None, None, 4), slice(None)),
    "encoder.stages.0.1.dwconv.weight": (slice(None, None, 4), slice(None)),
    "encoder.downsample_layers.1.1.weight": (slice(None, None, 8), slice(None, None, 8)),
    "proj.weight": (slice(None, None, 16), slice(None, None, 8)),
    "mask_token": (slice(None), slice(None, None, 4)),
    "loss_fn.log_vars": (slice(None),),
    "pred_dict.sentinel2.weight": (slice(None, None, 32), slice(None, None, 16)),
    "pred_dict.eco_region.weight": (slice(None, None, 32), slice(None, None, 16)),
}


def case_cfg(c):
    return make_cfg(c["model"], c["img"], c["patch"], out_modalities=M.subset(c["subset"]),
                    norm_pix_loss=c["norm_pix"], loss_aggr=c["aggr"], decoder_depth=c.get("decoder_depth", 1),
                    sparse=c.get("sparse", True), use_orig_stem=c.get("orig_stem", False))


def case_data(c, cfg):
    sd = make_state_dict(cfg, seed=c["wseed"])
    inputs, _ = make_inputs(cfg, c["N"], seed=c["iseed"])
    if c.get("zero_pix"):
        g = torch.Generator().manual_seed(c["iseed"] + 7)
        z = torch.rand(c["N"], 1, cfg.img_size, cfg.img_size, generator=g) < 0.08
        inputs["sentinel2"] = inputs["sentinel2"] * (~z)
    torch.manual_seed(c["nseed"])
    noise = torch.randn(c["N"], cfg.num_patches)
    return sd, inputs, noise


def checks(t: torch.Tensor):
    t = t.detach().double()
    return np.array([t.sum().item(), (t * t).sum().item(), t.abs().max().item()], dtype=np.float64)


def strided(t: torch.Tensor, step):
    return t.detach().reshape(-1)[::step].float().numpy().copy()


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
