"""Shared pieces of the engine modules: the launch-program option table, dtype tags, the ctypes pointer helper and the parameter dictionary."""
import contextlib
import os
import sys
import ctypes as C
import math
from collections import OrderedDict

import torch

from . import _lib
from ._lib import EPI, PRO
from .config import ModelCfg
from .synth import dense_aliases, flat_param_spec, param_view, state_dict_spec

F32, BF16 = 0, 1

# Launch-program options (A/B switches of the step's structure): defaults are the measured-best choices on MI355X. Override per
# engine with Engine(options={...}); the developer variable MPMAE_ENGINE_OPTS="name=value,..." is read HERE on the host side
# (the C library itself reads no environment, include/mpmae_hip.h mpmae_set_option).
ENGINE_OPTIONS = dict(
    down_grouped=1,         # LayerNorm writes the grouped operand of the 2x2/2 convolution
    rsc_small=1,            # chunked row-streaming kernels at C = 40 / 80 too
    grn_fold=1,             # GRN finalisation recomputed in the fused kernels' prologues
    rsc_pf=1,               # LDS-staged GRN vectors in the narrow kernels (needed by grn_fold)
    dzr=1,                  # dz never materialised at small C
    lanes=1,                # weight gradients on a side HIP stream
    heads_merged=1,         # one GEMM / weight gradient per head family
    stem_im2col=1,          # 3x3 stem convolution through a materialised im2col
    stem_fused=1,           # fused stem tail (patch 8)
    stem_front=1,           # ... and the 3x3 convolution in front of it in the same launch (stem_front_kernel), which also writes the im2col matrix of the weight gradient
    loss_multi=1,           # one launch per loss kind
    loss_rows=1,            # continuous pixel losses: row-band forward kernel
    loss_rows_bwd=1,        # ... and its gradient twin
    img_side=1,             # image-level head chain on the side lane
    tail_fold_group=1,      # the LayerNorm-gradient fold group that runs in order on the main lane (tail_main) as ONE launch, like the stem kernel's three folds (library: FOLD_GROUP >= 0)
    img_dgrad_side=1,       # round 5: the image-level heads' data-gradient GEMM (eight workgroups, pure latency) on the weight-gradient lane in front of the heads' weight gradients
    prep_side=1,            # weight staging of the forward on the side lane
    prep_late=1,            # the side lane runs activity + poolings FIRST and the weight staging behind them: the first stage-0 kernel (depthwise, fp32 taps) only waits for the poolings, the first staged weight is needed 50 us later
    front_side=1,           # ... followed there by the pixel-activity map and its poolings (main lane: mask -> im2col)
    z_free=1,               # fused blocks: z = GRN(gelu(h)) is not stored by the forward; pwconv2's weight gradient rebuilds it from h in its operand prologue
    z_free_maxc=40,         # ... up to this width (the prologue's GELU costs the weight-gradient lane 17 us per launch; the forward saves 27 us per block at C = 40, 13 at C = 80)
    fold_loss=1,            # data parallel: the scalar loss rides in the first gradient bucket's all-reduce (no collective of its own)
    proj_compact=1,         # proj as a plain NT GEMM on compact rows: the token kernel assembles the decoder input, its backward gathers the visible rows (no scatter / gather GEMM variants)
    zero_side=1,            # the step's zero fills (statistics, flat gradients, padded stem dW) on the side lane, ONE loss finalisation per step
    wgrad_late=1,           # pw2's weight gradient issued behind the block's second fused kernel (one main-lane event per block)
    ps_xcd_barrier=1,       # round 6: the persistent stage kernels' grid barrier in its XCD-hierarchical form (8 group counters + a top counter, MpmaePsArgs.sync_words = 640); 0 = the flat arrival counter
    ps=2,                   # persistent per-sample stage kernels (ps.cuh): bit 1 = (C, S) = (160, 2), bit 0 = (320, 1); one launch per stage. 2 since late round 4: with the decoder / head GEMMs on the vendor route the per-block kernels at stage 3 (2 blocks, M = 3584) measure 3.868-3.873 vs 3.892-3.897 ms for 3 in three interleaved pairs (a tie in round 3); 1: 4.04
    rsc=1,                  # chunked row-streaming kernels (rsc.cuh) at C = 160 / 320
    act_in_stem=1,          # the fused stem kernel writes the pixel-activity bytes itself (it computes them anyway): no activity launch, the first stage-0 op waits for nothing on the side lane, the poolings run behind the stem
    tail_main=-1,           # the LAST depthwise weight gradient of the backward (stage 0, block 0) and its folds in order on the MAIN lane behind the data gradient (1) or on the weight-gradient lane (0); -1 = by the lane's load: 0 where the stage-0 blocks carry their pointwise weight gradients inside the main-lane kernels (wg_fused: the lane has slack - atto 3.435 / 3.451 vs 3.462 / 3.467 ms, profiles/r06/option_sweep.txt), 1 otherwise (round 4: the lane ended 115 us after the main lane, 4.06 -> 4.02 ms; tiny 112/16 in round 6: 14.635 vs 14.70 ms, profiles/r06/tiny_option_sweep.txt); 2: the unfused pointwise pw1 weight gradient too
    dz_ring=16,             # depth of the dz / dh scratch ring (the pw1 weight gradient on the side lane reads dh); >= blocks of the net:
    ring=16,                # / of the dd / dx rings: no main-lane op ever waits for the side lane to release a scratch buffer (3 / 4: +60 us)
    rsn_maxc=192,           # largest C with the GRN application / its backward fused into the NARROW row-streaming kernels (beyond: tiled GEMMs + element-wise kernels; 384 on tiny 112/16: 16.64 vs 15.15 ms)
    dzr_maxc=80,            # largest C recomputing dz
    dw_group=1,             # ... and ONE launch for its depthwise weight gradients (mpmae_dwconv7_wgrad_group), from this stage index on (stage 0 stays per block: its weight gradients are the tail of the backward; 9 = never)
    ln_fold_defer=1,        # the LayerNorm gamma / beta gradient folds of the fused pointwise backward kernels leave the main lane: one mpmae_fold_group per stage on the weight-gradient lane
    grn_group=1,            # dense decoder blocks: GRN statistics + finalisation + application as ONE launch per direction (mpmae_grn_group_fwd / _bwd, rows of a sample in registers between the passes), gamma / beta gradient folds deferred to the side lane
    wgrad_group=1,          # ONE launch (+ one fold) for all pwconv1 / pwconv2 weight gradients of an encoder stage (mpmae_wgrad_group), issued behind the stage's data-gradient chain
    down_fused=1,           # round 6: the LayerNorm in front of a 2x2/2 downsample convolution computed in the epilogue of the stage's LAST fused pwconv2 kernel (MpmaeRsArgs.dn_*: stages 0 -> 1 and 1 -> 2; x-hat, rstd and the grouped affine output leave from the kernel that has the row in registers): no mpmae_ln_fwd_down launch, the stage output itself is never stored or re-read
    wg_fused=1,             # round 6: pwconv1's weight gradient of the stage-0 blocks (C = 40, dz recomputed) INSIDE the fused backward kernel (MpmaeRsArgs.wg_ws: U = dh^T x-hat and db1 per persistent workgroup, folded by mpmae_rs_wgrad_fold with the LayerNorm affine applied by linearity): dh is never stored (100 MB per block), the transpose-read product over dh and xn and its fold leave the weight-gradient lane, the forward does not store xn
    stats_wgrad=1,          # blocks that recompute dz (C <= dzr_maxc): the GRN backward statistics come from pwconv2's weight gradient - T = dout^T gelu(h) on the MAIN lane (mpmae_rs which = 6, csrc/rst.cuh: one read of dout and h at the price of the statistics-only pass it replaces, 33.9 vs 34.7 us at stage 0), then mpmae_grn_stats_from_wgrad -> S0, S1, dW2, db2: the weight-gradient lane loses pwconv2's transpose-read product and fold over the same two tensors (72 + 12 us per stage-0 block), the forward never stores z at those widths. Round 6: 3.471 vs 3.569 ms (profiles/r06/ab_stats_wgrad.txt). 0 = statistics pass + separate weight gradient (round 5's route to the same T - the generic gemm_tn2 kernel on the main lane - measured 3.65 ms, slower than off, and is removed)
    grn_apply_fin=1,        # unfused sparse blocks (C = 320: tiled GEMMs + element-wise GRN passes): the GRN finalisation runs in the prologue of the element-wise pass (mpmae_grn_apply_fin / _bwd_apply_fin) - two launches fewer per block on the main lane
    loss_onepass=1,         # pixel losses in ONE pass (round 5): the forward kernels also write the loss gradient without its per-modality scalar; the scalar is folded into the heads' data-gradient weights (mpmae_head_scale) and weight-gradient fold (rowscale): the dloss:pix_* kernels (69 us of the main lane, a second pass over predictions and targets) leave the step
    det=0,                  # 1 = reproducible forward: no persistent stage kernel (its GRN exchange is float atomics), library option DET = 1 (every fold as one ordered row group - parameter-gradient folds included); 4.53-4.55 vs 3.89-3.90 ms
)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _rup(x, m):
    return (x + m - 1) // m * m


class _ParamDict(OrderedDict):
    """Parameters (or gradients) by state-dict key; `alias` holds engine-internal names of the same storage (dense encoder: the stem and
    downsampling tensors under the sparse encoder's names and layouts, synth.dense_aliases) - looked up, never iterated."""

    def __init__(self):
        super().__init__()
        self.alias = {}

    def __missing__(self, key):
        return self.alias[key]
